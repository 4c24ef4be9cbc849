"""`src.models.motionencoder.pytorch_i3d.InceptionI3d` is the reference's FVD feature extractor (a Kinetics-400 Inception-3D).
It is outside the hot path and useless without its pretrained weights, which cannot be obtained offline:
the name exists so that the reference's configs/model/evaluator.yaml composes, and building it says what to do instead."""


class InceptionI3d:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "InceptionI3d is not built (pretrained I3D weights are not obtainable offline).  Keep model.do_evaluation=false, or "
            "point model.evaluator.videoencoder._target_ at a feature extractor you have weights for: src.utils.evaluator.Evaluator "
            "accepts any module mapping (B,3,T,H,W) clips to (B,F) features")
