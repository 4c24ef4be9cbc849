"""Reference checkpoint interop: load the reference's stage-1 / stage-2 checkpoints into the
HIP-path modules and write checkpoints the reference can read back.

Layouts handled (reference: src/models/multistage_text_motion_model.py:113-122 strips the first 10 characters of every key,
i.e. the Lightning attribute prefix "generator."; src/tasks/train_task.py:70 saves Lightning checkpoints):
  * a bare ``state_dict`` of the module (VQVAE or DiffusionTransformer / DiscreteDiffusion);
  * a Lightning checkpoint ``{"state_dict": {"generator.<key>": ..., "autoencoder.<key>": ...}, ...}``.
Keys that exist in the reference but carry no state here are dropped explicitly (never silently): the causal ``mask``
buffers CrossAttention registers and never uses (transformer_utils.py:91-93), and the cached ``zero_vector``
(diffusion_transformer.py:233-234), and the text tower under ``textencoder.`` whose output the reference zeroes.
Anything else unexpected or missing, or a shape mismatch, raises.

Files are read with ``torch.load(..., weights_only=True)``: nothing in a checkpoint is executed."""
import re

import torch

DEAD_KEYS = (re.compile(r"(^|\.)attn2\.mask$"), re.compile(r"(^|\.)zero_vector$"),
             # the text tower: the reference multiplies its output by zero (discrete_diffusion.py:25,49), so its weights (CLIP
             # ViT-B/32 in the reference, a deterministic stand-in here) carry no state of the path
             re.compile(r"^textencoder\."))
PREFIXES = ("generator.", "autoencoder.")


def _is_dead(key):
    return any(p.search(key) for p in DEAD_KEYS)


def read_state(path_or_state):
    """-> flat {key: tensor} from a path, a Lightning checkpoint dict or a bare state_dict."""
    state = path_or_state
    if not isinstance(state, dict):
        state = torch.load(state, map_location="cpu", weights_only=True)
    if "state_dict" in state and isinstance(state["state_dict"], dict):
        state = state["state_dict"]
    return {k: v for k, v in state.items() if torch.is_tensor(v)}


def select(state, prefix):
    """Entries under `prefix` with the prefix removed (the reference's `param_key[10:]`); the whole dict if no key has it."""
    if prefix and any(k.startswith(prefix) for k in state):
        return {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
    return dict(state)


def load_reference_checkpoint(module, path_or_state, prefix="auto"):
    """Load into `module` (strictly, apart from the documented dead keys).  prefix: "generator." / "autoencoder." / None /
    "auto" (use the one prefix under which the module's keys are found).  Returns the list of dropped dead keys."""
    state = read_state(path_or_state)
    own = module.state_dict()
    if prefix == "auto":
        prefix = None
        for cand in PREFIXES:
            if any(k.startswith(cand) for k in state) and any(k in own for k in select(state, cand)):
                prefix = cand
                break
    sub = select(state, prefix)
    dropped = sorted(k for k in sub if k not in own and _is_dead(k))
    sub = {k: v for k, v in sub.items() if k not in dropped}
    unexpected = sorted(k for k in sub if k not in own)
    missing = sorted(k for k in own if k not in sub and not _is_dead(k))
    if unexpected or missing:
        raise KeyError(f"checkpoint does not match {type(module).__name__}: unexpected {unexpected[:5]}"
                       f"{'...' if len(unexpected) > 5 else ''}, missing {missing[:5]}{'...' if len(missing) > 5 else ''}")
    for k, v in sub.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != module shape {tuple(own[k].shape)}")
    module.load_state_dict(sub, strict=False)
    for m in module.modules():                     # packed-weight caches of the HIP shells
        if hasattr(m, "_packed"):
            m._packed = None
    cb = getattr(module, "codebook", None)
    if cb is not None and hasattr(cb, "_need_init"):
        cb._need_init = False                      # a trained codebook must not be re-initialised from data
    return dropped


def lightning_state(generator=None, autoencoder=None):
    """A Lightning-layout state_dict with the attribute prefixes "generator." / "autoencoder.".

    What the reference can read back: its load_checkpoints (multistage_text_motion_model.py:113-122) strips exactly 10
    characters -- len("generator.") -- from EVERY key of whichever file it opens.  So a VQ-VAE meant for the reference's stage 2
    must be exported the way its own stage 1 saves it, as `lightning_state(generator=vqvae)` (TextMotionModel holds the VQ-VAE as
    `self.generator`); keys written under "autoencoder." (12 characters) would come out mangled there.  A stage-2 file
    (`generator=<DiscreteDiffusion>, autoencoder=<VQVAE>`) is this build's own resume format and loads through
    load_reference_checkpoint's prefix selection."""
    out = {}
    for prefix, mod in (("generator.", generator), ("autoencoder.", autoencoder)):
        if mod is not None:
            out.update({prefix + k: v.detach().cpu() for k, v in mod.state_dict().items()})
    return {"state_dict": out}
