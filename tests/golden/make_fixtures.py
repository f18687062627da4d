"""Regenerates tests/golden/ from the reference's own test DATA (run in the build container only;
/root/reference does not exist on the GPU box and nothing at test time reads it).

Fixtures are data, not source:
  * the three PCM16 WAV inputs the reference's tests read (test_data/*.wav), byte for byte;
  * the three Matlab matrices of speedy_test.cc:859-871 (tapestry_*_data.txt), re-encoded as float32
    .npz (values parsed with numpy, same as the reference's `ss >> value` into float).
"""
import os
import shutil
import sys

import numpy as np

REF = "/root/reference/test_data"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        sys.exit("reference test_data not present; fixtures are committed, nothing to do")
    for name in ("tapestry.wav", "tapestry22050.wav", "negative_speed.wav"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(HERE, name))
    mats = {}
    for key, name in (("spectrogram", "tapestry_spectrogram_data.txt"),
                      ("normalized", "tapestry_normalized_spectrogram_data.txt"),
                      ("features", "tapestry_features_data.txt")):
        mats[key] = np.loadtxt(os.path.join(REF, name), comments="#", dtype=np.float64).astype(np.float32)
        print(key, mats[key].shape)
    np.savez_compressed(os.path.join(HERE, "tapestry22050_matlab.npz"), **mats)


if __name__ == "__main__":
    main()
