"""fp64 operations per analysis frame, counted from this repo's DFT spec (oracle/orc_speedy.c orc_butterfly /
orc_plan_execute / orc_specplan_run / orc_log restate it operation for operation) -- the numerator of bench.py's
`roofline.valu_fp64`.  Counted: IEEE double additions / subtractions, multiplications, divisions and square roots that the
spec performs on values that are not structural constants: multiplications by the twiddle 1 (output 0 of every butterfly,
the whole last stage, butterfly p = 0) are not counted (the kernel skips them, bit-identically), nor are the operations of
the first stage on the zero padding of the packed frame.  A division and a square root count as ONE operation each (the
hardware runs ~10 / ~15 instructions for them: reported separately).  Everything else of a frame (mono mix, pre-emphasis,
window, energy, gate, the float quotient in front of the log, the float accumulation) is fp32 or integer and not counted.

    python tools/flop_count.py            # prints the table, writes profiles/flop_model.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def radices(n):
    r = []
    while n % 4 == 0:
        r.append(4); n //= 4
    while n % 2 == 0:
        r.append(2); n //= 2
    while n % 3 == 0:
        r.append(3); n //= 3
    while n % 5 == 0:
        r.append(5); n //= 5
    p = 7
    while n > 1:
        while n % p == 0:
            r.append(p); n //= p
        p += 2
    return r


def butterfly(r):
    """(adds, muls) of one radix-r butterfly of the spec (orc_butterfly)."""
    if r == 2:
        return 4, 0
    if r == 4:
        return 16, 0
    if r == 3:
        return 12, 4
    if r == 5:
        return 32, 16
    h = (r - 1) // 2
    return 6 * h + h * (4 * h + 2), 4 * h * h


def dft(W):
    """The W-point complex transform of the packed frame z[n] = x[2n] + i x[2n+1] (upper half zero)."""
    adds = muls = 0
    s, cur = 1, W
    rs = radices(W)
    for st, r in enumerate(rs):
        m = cur // r
        ba, bm = butterfly(r)
        n_bf = m * s
        if st == 0 and r == 4:
            # inputs 2 and 3 of every first-stage butterfly are the zero padding: t0 = t1 = a0, t2 = t3 = a1 -- the
            # four outputs are a0 + a1, a0 - i a1, a0 - a1, a0 + i a1: 8 additions instead of 16
            ba = 8
        adds += n_bf * ba
        muls += n_bf * bm
        if st < len(rs) - 1:                       # the last stage's twiddles are all 1 (p = 0 only)
            nontrivial = (m - 1) * s * (r - 1)     # p >= 1, j >= 1
            adds += 2 * nontrivial
            muls += 4 * nontrivial
        s *= r
        cur = m
    return adds, muls


def frame(W):
    a, m = dft(W)
    # untangle + magnitude per bin k < W (orc_specplan_run): er, ei (2 add, 2 mul); dr, di (2 add); o (2 mul);
    # xr, xi (4 mul, 4 add); |X| = sqrt(xr^2 + xi^2) (2 mul, 1 add, 1 sqrt); bin W: one subtraction
    ua, um, usq = W * (2 + 2 + 4 + 1) + 1, W * (2 + 2 + 4 + 2), W
    # |log| of the 239 (W - 1) ratios a frame can have (orc_log, the fdlibm polynomial): f = x - 1; s = f / (2 + f);
    # z = s s; w = z z; t1 (3 mul 2 add); t2 (4 mul 3 add); R; hfsq = 0.5 f f; dk ln2_hi - ((hfsq - (s (hfsq + R) + dk ln2_lo)) - f)
    la, lm, ld = 13, 14, 1
    bins = W - 1
    return {"W": W, "radices": radices(W),
            "dft": {"add": a, "mul": m},
            "untangle_magnitude": {"add": ua, "mul": um, "sqrt": usq},
            "log_terms": {"bins": bins, "add": bins * la, "mul": bins * lm, "div": bins * ld},
            "flop_per_frame": a + m + ua + um + usq + bins * (la + lm + ld),
            "of_which_div_sqrt": usq + bins * ld}


def main():
    model = {"_note": "tools/flop_count.py: fp64 operations per analysis frame from the DFT spec (DESIGN.md 4, 6); a division / "
                      "square root counts as one operation",
             "16000": frame(240), "22050": frame(330)}
    for k in ("16000", "22050"):
        f = model[k]
        print("%s Hz: W = %d = %s: DFT %d add + %d mul, untangle + magnitude %d + %d + %d sqrt, log terms %d x (13 add + 14 mul + 1 div)"
              " = %d fp64 operations per frame (%d of them div / sqrt)" % (
                  k, f["W"], " x ".join(str(r) for r in f["radices"]), f["dft"]["add"], f["dft"]["mul"],
                  f["untangle_magnitude"]["add"], f["untangle_magnitude"]["mul"], f["untangle_magnitude"]["sqrt"],
                  f["log_terms"]["bins"], f["flop_per_frame"], f["of_which_div_sqrt"]))
    out = os.path.join(ROOT, "profiles", "flop_model.json")
    if "--no-write" not in sys.argv:
        with open(out, "w") as fh:
            json.dump(model, fh, indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
