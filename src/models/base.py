"""Hook plumbing of the reference's BaseModel (src/models/base.py:27-63) without a hard Lightning dependency."""
import torch.nn as nn

try:                                            # use Lightning when the environment has it
    from pytorch_lightning import LightningModule as _Base
except Exception:                               # pragma: no cover - not installed in this image
    class _Base(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, d, **k):
            self._logged = dict(d)

        @property
        def device(self):
            return next(self.parameters()).device


class BaseModel(_Base):
    def training_step(self, batch, batch_idx):
        return self.allsplit_step("train", batch, batch_idx)

    def validation_step(self, batch, batch_idx):
        return self.allsplit_step("val", batch, batch_idx)

    def test_step(self, batch, batch_idx):
        return self.allsplit_step("test", batch, batch_idx)
