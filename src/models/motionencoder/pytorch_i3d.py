"""`src.models.motionencoder.pytorch_i3d.InceptionI3d` (the FVD evaluator's feature extractor, configs/model/evaluator.yaml) ->
HIP-backed drop-in with the reference's constructor, methods and state_dict keys."""
from gsdd_amd.i3d import InceptionI3d, InceptionModule, MaxPool3dSamePadding, Unit3D  # noqa: F401
