"""ctypes loader of oracle/stickman_oracle.c (TEST INFRASTRUCTURE ONLY).  build() compiles it with gcc."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "stickman_oracle.c")
LIB = os.path.join(HERE, "_build", "libstickman_oracle.so")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-std=c99", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


def raster(kps: np.ndarray, body, cmds, h: int, w: int, thickness: int = 1) -> np.ndarray:
    """kps [B, J, 2] float32 -> uint8 [B, 3, H, W] by the sequential OpenCV-4.1.2-style restatement; ``thickness``: the
    cv2.line thickness of every line of the frame (lib/utils.py:334-339)."""
    lib = ctypes.CDLL(build())
    kps = np.ascontiguousarray(kps, dtype=np.float32)
    body = np.ascontiguousarray(body, dtype=np.int32)
    cmds = np.ascontiguousarray(cmds, dtype=np.int32).reshape(-1, 6)
    b, j = kps.shape[:2]
    out = np.zeros((b, 3, h, w), dtype=np.uint8)
    lib.stickman_raster_oracle_thick(kps.ctypes.data_as(ctypes.c_void_p), b, j, body.ctypes.data_as(ctypes.c_void_p),
                                     len(body), cmds.ctypes.data_as(ctypes.c_void_p), len(cmds),
                                     out.ctypes.data_as(ctypes.c_void_p), h, w, int(thickness))
    return out
