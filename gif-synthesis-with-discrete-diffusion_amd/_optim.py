"""Host-side helpers shared by the two trainers: a zero-initialised gradient arena (one fill per step instead of one per tensor)
and Adam over all parameters in one launch (gsdd_adam_multi)."""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr

ADAM_CHUNK = 4096


class GradArena:
    """Hands out zero-filled views of one flat fp32 buffer.  `reset()` zeroes the buffer (one kernel); a request that does not fit
    falls back to torch.zeros and grows the buffer for the next step."""

    def __init__(self, device, capacity=0):
        self.device, self.buf, self.off, self.want = device, None, 0, int(capacity)
        self.spilled = False               # a request of this step did not fit (its tensor lives outside the buffer)

    def reset(self):
        self.spilled = False
        if self.buf is None or self.buf.numel() < self.want:
            self.buf = torch.zeros((self.want,), dtype=torch.float32, device=self.device)
        else:
            self.buf.zero_()
        self.off = 0
        self.want = 0

    def zeros(self, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        n4 = (n + 3) & ~3                                  # keep every view 16-byte aligned (float4 kernels)
        self.want += n4
        if self.buf is not None and self.off + n4 <= self.buf.numel():
            v = self.buf[self.off:self.off + n].view(shape)
            self.off += n4
            return v
        self.spilled = True
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def zeros_like(self, t):
        return self.zeros(tuple(t.shape))


class MultiAdam:
    """Adam state (m, v flat) for a fixed list of parameters; `step(grads)` = one gsdd_adam_multi launch."""

    def __init__(self, params, lr, betas, eps):
        self.params = list(params)                                        # [(name, tensor)]
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        dev = self.params[0][1].device
        sizes = [p.numel() for _, p in self.params]
        self.offs = np.concatenate([[0], np.cumsum([(n + 3) & ~3 for n in sizes])]).astype(np.int64)
        self.m = torch.zeros((int(self.offs[-1]),), dtype=torch.float32, device=dev)
        self.v = torch.zeros_like(self.m)
        self._key, self._table = None, None

    def state(self):
        return {"m": self.m.detach().cpu(), "v": self.v.detach().cpu(), "step": int(self.step_count)}

    def load_state(self, st):
        if st is None:
            return
        if st["m"].numel() != self.m.numel():
            raise ValueError(f"Adam state holds {st['m'].numel()} values, this parameter list needs {self.m.numel()}")
        self.m.copy_(st["m"])
        self.v.copy_(st["v"])
        self.step_count = int(st["step"])

    def _build_table(self, grads):
        rows = []
        mb, vb = self.m.data_ptr(), self.v.data_ptr()
        for (name, p), off in zip(self.params, self.offs[:-1]):
            n = p.numel()
            starts = np.arange(0, n, ADAM_CHUNK, dtype=np.int64)
            t = np.empty((len(starts), 5), dtype=np.int64)
            t[:, 0] = p.data_ptr() + 4 * starts
            t[:, 1] = grads[name].data_ptr() + 4 * starts
            t[:, 2] = mb + 4 * (off + starts)
            t[:, 3] = vb + 4 * (off + starts)
            t[:, 4] = np.minimum(ADAM_CHUNK, n - starts)
            rows.append(t)
        return np.concatenate(rows)

    def step(self, grads, stream=None, step_dev=None):
        """step_dev: int64[1] device tensor holding the step count of THIS update (>= 1): the launch then bakes nothing that changes
        from step to step (captured training step); the host-side count still advances so that checkpoints see it."""
        for name, p in self.params:
            g = grads[name]
            if not g.is_contiguous() or g.shape != p.shape:
                grads[name] = g.contiguous().view(p.shape)
        key = tuple(grads[name].data_ptr() for name, _ in self.params) + tuple(p.data_ptr() for _, p in self.params)
        if key != self._key:
            self._table = torch.from_numpy(self._build_table(grads)).to(self.params[0][1].device)
            self._key = key
        self.step_count += 1
        if step_dev is not None:
            check(lib().gsdd_adam_multi_dev(ptr(self._table), self._table.shape[0], self.lr, self.betas[0], self.betas[1], self.eps,
                                            ptr(step_dev), stream_ptr(stream)))
            return
        check(lib().gsdd_adam_multi(ptr(self._table), self._table.shape[0], self.lr, self.betas[0], self.betas[1], self.eps,
                                    self.step_count, stream_ptr(stream)))
