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


def butterfly(r, v2):
    """(adds, muls, fmas) of one radix-r butterfly of the spec (orc_butterfly / orc_butterfly_v2)."""
    if r == 2:
        return 4, 0, 0
    if r == 4:
        return 16, 0, 0
    h = (r - 1) // 2
    if not v2:
        if r == 3:
            return 12, 4, 0
        if r == 5:
            return 32, 16, 0
        return 6 * h + h * (4 * h + 2), 4 * h * h, 0
    if r == 3:
        return 10, 2, 2          # t2 = fma(-1/2, t1, a0)
    if r == 5:
        return 20, 4, 12         # m1, m2: two fma per component; n1, n2: one product and one fma per component
    return 10 * h, 2 * h, h * (4 * h - 2)


def dft(W, v2):
    """The W-point complex transform of the packed frame z[n] = x[2n] + i x[2n+1] (upper half zero)."""
    adds = muls = fmas = 0
    s, cur = 1, W
    rs = radices(W)
    for st, r in enumerate(rs):
        m = cur // r
        ba, bm, bf = butterfly(r, v2)
        n_bf = m * s
        if st == 0 and r == 4:
            # inputs 2 and 3 of every first-stage butterfly are the zero padding: t0 = t1 = a0, t2 = t3 = a1 -- the
            # four outputs are a0 + a1, a0 - i a1, a0 - a1, a0 + i a1: 8 additions instead of 16
            ba = 8
        adds += n_bf * ba
        muls += n_bf * bm
        fmas += n_bf * bf
        if st < len(rs) - 1:                       # the last stage's twiddles are all 1 (p = 0 only)
            nontrivial = (m - 1) * s * (r - 1)     # p >= 1, j >= 1
            if v2:                                 # re = fma(br, wx, -(bi wy)), im = fma(br, wy, bi wx)
                muls += 2 * nontrivial
                fmas += 2 * nontrivial
            else:
                adds += 2 * nontrivial
                muls += 4 * nontrivial
        s *= r
        cur = m
    return adds, muls, fmas


def frame(W, v2=True):
    a, m, f = dft(W, v2)
    bins = W - 1
    if v2:
        # untangle + magnitude per bin (orc_specplan_run, spec v2): dr, di, ar + br, ai + bi (4 add); 2 xr, 2 xi (4 fma);
        # |X| = sqrt(fma(2xr, 2xr, 2xi 2xi) / 4) (2 mul, 1 fma, 1 sqrt); bin W: one subtraction
        ua, um, uf, usq = W * 4 + 1, W * 2, W * 5, W
        # |log| of the W - 1 ratios a frame can have, log spec v2 (orc_log_v2_f32): r, w (2 fma); hi, lo (3 add); lo + logc_lo (1 add);
        # fma(k, Ln2lo, .); r r (1 mul); five fma of the polynomial; fma(r r, p, lo) + hi (1 fma, 1 add)
        la, lm, lf, ld = 5, 1, 9, 0
    else:
        ua, um, uf, usq = W * (2 + 2 + 4 + 1) + 1, W * (2 + 2 + 4 + 2), 0, W
        la, lm, lf, ld = 13, 14, 0, 1
    ops = a + m + f + ua + um + uf + usq + bins * (la + lm + lf + ld)
    fma_total = f + uf + bins * lf
    return {"W": W, "radices": radices(W), "spec": "v2 (round 5: fused multiply-adds, table-driven log)" if v2 else "v1 (rounds 1-4: no fma, fdlibm log)",
            "dft": {"add": a, "mul": m, "fma": f},
            "untangle_magnitude": {"add": ua, "mul": um, "fma": uf, "sqrt": usq},
            "log_terms": {"bins": bins, "add": bins * la, "mul": bins * lm, "fma": bins * lf, "div": bins * ld},
            "flop_per_frame": ops,                         # operations = instructions' worth: an fma is ONE
            "flops_counting_fma_twice": ops + fma_total,
            "of_which_fma": fma_total,
            "of_which_div_sqrt": usq + bins * ld}


def main():
    model = {"_note": "tools/flop_count.py: fp64 OPERATIONS per analysis frame from the DFT and log specs (DESIGN.md 4, 4a, 6); an fma, a "
                      "division and a square root count as one operation each (flops_counting_fma_twice: the usual flop count)",
             "16000": frame(240), "22050": frame(330), "v1": {"16000": frame(240, False), "22050": frame(330, False)}}
    for k in ("16000", "22050"):
        f = model[k]
        print("%s Hz: W = %d = %s: DFT %d add + %d mul + %d fma, untangle + magnitude %d + %d + %d fma + %d sqrt, log terms %d x 15"
              " = %d fp64 operations per frame (%d fma, %d sqrt; spec v1: %d)" % (
                  k, f["W"], " x ".join(str(r) for r in f["radices"]), f["dft"]["add"], f["dft"]["mul"], f["dft"]["fma"],
                  f["untangle_magnitude"]["add"], f["untangle_magnitude"]["mul"], f["untangle_magnitude"]["fma"], f["untangle_magnitude"]["sqrt"],
                  f["log_terms"]["bins"], f["flop_per_frame"], f["of_which_fma"], f["of_which_div_sqrt"], model["v1"][k]["flop_per_frame"]))
    out = os.path.join(ROOT, "profiles", "flop_model.json")
    if "--no-write" not in sys.argv:
        with open(out, "w") as fh:
            json.dump(model, fh, indent=1)
        print("wrote", out)


if __name__ == "__main__":
    main()
