from .blocks import BatchRNN, InferenceBatchSoftmax, Lookahead, MaskConv, SequenceWise
from .deepspeech import DeepSpeech
