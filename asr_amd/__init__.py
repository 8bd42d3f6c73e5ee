"""asr_amd — MI355X-native (gfx950) DeepSpeech2 training hot path, drop-in for the model + loss of
zakuro-ai/asr (`asr_deepspeech`).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"

from .ctc import CTCLoss
from .functional import _collate_fn, check_loss, reduce_tensor, to_np
from .modules import DeepSpeech
from .optim import FusedAdamW
from .vars import resolve_rnn_type, supported_rnns, supported_rnns_inv
from .device import autocast, make_grad_scaler, resolve_device


def seed_like_reference(seed: int = 123456):
    """The reference seeds torch at import time (asr_deepspeech/vars.py:13-15); here it is explicit."""
    import torch
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
