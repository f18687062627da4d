"""The resource numbers the engine's launch-mode decision is fed (spx_debug_mode_resources) for the batch shapes the tests and the
bench rely on -> profiles/kernel_resources.json.  Runs on the GPU box (the register counts come from hipFuncGetAttributes);
tests/test_mode_table.py replays the decision on the CPU from this file, tests/test_gpu_parity.py checks the file against the library.
Usage: python tools/kernel_resources.py [OUT.json]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = ["cu_count", "lds_per_cu", "walk_lds", "walk_waves", "walk_vgprs", "walk_fast", "walk_nwc", "lean_lds", "lean_waves", "lean_vgprs",
          "lean_fast", "lean_nwc", "lean_valid", "tension_lds", "tension_vgprs", "tile_default", "tile_big", "tile_small", "an_lds_default",
          "an_lds_small", "an_vgprs_default", "an_vgprs_small"]
# (sample rate, channels, streams, every job speeds up)
SHAPES = [(16000, 1, 256, 1), (16000, 1, 97, 1), (16000, 1, 64, 1), (16000, 2, 256, 1), (16000, 2, 128, 1), (22050, 1, 256, 1),
          (22050, 2, 256, 1), (22050, 2, 128, 1), (44100, 1, 256, 1), (48000, 2, 256, 1), (48000, 1, 64, 1), (16000, 1, 512, 1),
          (16000, 1, 600, 1), (16000, 1, 1024, 1), (16000, 1, 2048, 1), (16000, 1, 256, 0), (8000, 1, 256, 1), (11025, 1, 256, 1),
          (24000, 1, 256, 1), (32000, 1, 256, 1)]


def collect():
    from speedy_amd._lib import lib
    L = lib()
    out = (C.c_longlong * 22)()
    shapes = {}
    for rate, ch, n, sp in SHAPES:
        if L.spx_debug_mode_resources(rate, ch, n, sp, out) != 0:
            raise RuntimeError("spx_debug_mode_resources failed for %r" % ((rate, ch, n, sp),))
        shapes["%d,%d,%d,%d" % (rate, ch, n, sp)] = [int(v) for v in out]
    return {"fields": FIELDS, "shapes": shapes,
            "note": "spx_debug_mode_resources per (sample rate, channels, streams, speed-up only): what spx_choose_mode (speedy_amd/csrc/spx_mode.h) "
                    "is fed on this device; tools/kernel_resources.py"}


if __name__ == "__main__":
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.json")
    json.dump(collect(), open(dst, "w"), indent=1)
    print("wrote", dst)
