"""Helpers shared by the -m gpu parity tests: HIP product path vs the CPU oracle / golden vectors."""
import numpy as np
import torch

MASK32 = 0xFFFFFFFF


def hash_u32(x: np.ndarray) -> np.ndarray:
    """Bit-for-bit restatement of vunet_hash_u32 (csrc/common.h)."""
    x = x.astype(np.uint64) & MASK32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & MASK32
    x ^= x >> 15
    x = (x * 0x846CA68B) & MASK32
    x ^= x >> 16
    return x


def dropout_keep_mask(shape, p: float, seed: int) -> torch.Tensor:
    """CPU restatement of the conv prologue's keep-mask: keep iff hash(idx + seed) >= p * 2^32."""
    n = int(np.prod(shape))
    if p <= 0:
        return torch.ones(shape)
    thresh = min(int(p * 4294967296.0), 0xFFFFFFFF) or 1
    idx = (np.arange(n, dtype=np.uint64) + (seed & MASK32)) & MASK32
    keep = hash_u32(idx) >= thresh
    return torch.from_numpy(keep.astype(np.float32)).reshape(shape)


def assert_close(actual, expected, rtol=1e-4, atol=1e-4, name=""):
    a = actual.detach().float().cpu().numpy() if isinstance(actual, torch.Tensor) else np.asarray(actual)
    e = expected.detach().float().cpu().numpy() if isinstance(expected, torch.Tensor) else np.asarray(expected)
    assert a.shape == e.shape, f"{name}: shape {a.shape} vs {e.shape}"
    err = np.abs(a - e)
    tol = atol + rtol * np.abs(e)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{name}: max abs err {err.max():.3e} (at {i}: {a[i]} vs {e[i]}), "
                             f"{(err > tol).mean() * 100:.3f}% outside rtol={rtol} atol={atol}")


def psnr(a: torch.Tensor, b: torch.Tensor, peak: float = 2.0) -> float:
    mse = float(((a.double().cpu() - b.double().cpu()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * np.log10(peak * peak / mse)
