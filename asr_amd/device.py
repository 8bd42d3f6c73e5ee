"""Device resolution (mirror of asr_deepspeech/device.py:8-46).  ROCm torch reports MI355X as 'cuda'."""
import warnings

import torch


def resolve_device(spec="auto") -> torch.device:
    if spec is None:
        spec = "auto"
    if isinstance(spec, torch.device):
        return spec
    key = str(spec).lower()
    if key == "cpu":
        return torch.device("cpu")
    if key in ("cuda", "gpu"):
        if torch.cuda.is_available():
            return torch.device("cuda")
        warnings.warn("CUDA device requested but no GPU is available; falling back to CPU.", UserWarning, stacklevel=2)
        return torch.device("cpu")
    if key == "auto":
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")
    raise ValueError(f"Unknown device spec: {spec!r}")


def make_grad_scaler(device, enabled: bool = True) -> "torch.amp.GradScaler":
    """Kept for API parity (device.py:37-40).  The HIP path computes in fp32 (bf16 MFMA operands are
    a build-side option, never fp16), so loss scaling is an identity and the scaler stays disabled."""
    dev = resolve_device(device)
    return torch.amp.GradScaler(dev.type, enabled=False)


def autocast(device, enabled: bool = True):
    """API parity (device.py:43-46): a no-op context — the kernels pick their own operand precision."""
    dev = resolve_device(device)
    return torch.amp.autocast(device_type=dev.type, enabled=False)
