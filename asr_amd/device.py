"""Device resolution (mirror of asr_deepspeech/device.py:8-46).  ROCm torch reports MI355X as 'cuda'."""
import warnings

import torch


def _gpu_or_cpu(warn: bool) -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda")
    if warn:
        warnings.warn("CUDA device requested but no GPU is available; falling back to CPU.", UserWarning, stacklevel=3)
    return torch.device("cpu")


# spelling -> how to resolve it.  Same accepted spellings, warning and error text as the reference's resolver (its
# tests/test_device.py pins them): "cuda"/"gpu" degrade to the CPU with a warning, "auto"/None pick the GPU when there is one.
_DEVICE_SPECS = {
    "cpu": lambda: torch.device("cpu"),
    "cuda": lambda: _gpu_or_cpu(warn=True),
    "gpu": lambda: _gpu_or_cpu(warn=True),
    "auto": lambda: _gpu_or_cpu(warn=False),
}


def resolve_device(spec="auto") -> torch.device:
    if isinstance(spec, torch.device):
        return spec
    resolver = _DEVICE_SPECS.get("auto" if spec is None else str(spec).lower())
    if resolver is None:
        raise ValueError(f"Unknown device spec: {spec!r}")
    return resolver()


def make_grad_scaler(device, enabled: bool = True) -> "torch.amp.GradScaler":
    """Kept for API parity (device.py:37-40).  The HIP path computes in fp32 (bf16 MFMA operands are
    a build-side option, never fp16), so loss scaling is an identity and the scaler stays disabled."""
    dev = resolve_device(device)
    return torch.amp.GradScaler(dev.type, enabled=False)


def autocast(device, enabled: bool = True):
    """API parity (device.py:43-46): a no-op context — the kernels pick their own operand precision."""
    dev = resolve_device(device)
    return torch.amp.autocast(device_type=dev.type, enabled=False)
