"""`src.datamodules.msrvtt_datamodule.MSRVTTDataModule` (reference: src/datamodules/msrvtt_datamodule.py; configs/datamodule/
msrvtt.yaml): captioned clips for the text-conditioned workload C5.  Same two sources as UCF101DataModule; in a clip folder the
directory name is the caption."""
from src.datamodules.ucf101_datamodule import UCF101DataModule


class MSRVTTDataModule(UCF101DataModule):
    caption = "caption"
