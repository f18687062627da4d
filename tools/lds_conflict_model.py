"""Developer tool: LDS bank-conflict model of the walk kernel's per-step LDS instructions (16 kHz mono, 4 search waves),
after MI355X_MICROARCH.md's LDS section: a wave64 ds_read_b32 is served in two groups of 32 lanes, bank = (addr / 4) mod 32,
one LDS cycle per group plus one per extra DISTINCT address on a busy bank (identical addresses broadcast); ds_read2_b32 =
two such accesses; an atomic ds_add serialises lanes that hit the same address.  Prints LDS cycles per step and per phase,
averaged over window offsets and coarse winners, for a given padding of the shifted copies.
Usage: python tools/lds_conflict_model.py [pad_mono] [pad_pl] [wcap]"""
import sys
from collections import defaultdict

pad_mono = int(sys.argv[1]) if len(sys.argv) > 1 else 48
pad_pl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wcap = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
rate, skip = 16000, 4
minP, maxP = rate // 400, rate // 65
minC, nC = minP // skip, maxP // skip - minP // skip + 1
NWM = int(sys.argv[4]) if len(sys.argv) > 4 else 4
DEAL = sys.argv[5] if len(sys.argv) > 5 else "spread"   # coarse dealing: "block" = task = thread (round 3), "spread" = consecutive tasks on different waves (round 4)
FCMD_INTS = 64
# layout (fast_lds_layout_i)
o = 0
off_cmd = o; o += 2 * FCMD_INTS * 4
o += 16
off_sumC = o; o += 512
off_sumR = o; o += 512
o += 512
o += ((maxP + 2) * 8 + 15) & ~15
mb = ((wcap + 8) * 2 + 15) & ~15
off_mono = o; o += mb + pad_mono
off_monoB = o; o += mb
plStride = ((wcap // skip + 4) + 1) & ~1
plStrideB = plStride * 2
plb = (plStride * skip * 2 + 15) & ~15
off_pl = o; o += plb + pad_pl
off_plB = o; o += plb
dA = off_monoB - 2 - off_mono
dPl = off_plB - 2 - off_pl


def pair_addr(base, d2, e):
    return base + 2 * e + (e & 1) * d2


def read_cycles(addrs):
    """addrs: list of 64 byte addresses or None (inactive lane).  LDS cycles of one ds_read_b32."""
    cyc = 0
    for g in range(2):
        banks = defaultdict(set)
        for a in addrs[32 * g:32 * g + 32]:
            if a is not None:
                banks[(a // 4) % 32].add(a // 4)
        cyc += max([len(v) for v in banks.values()] + [1])
    return cyc


def atomic_cycles(addrs):
    cyc = 0
    for g in range(2):
        banks = defaultdict(list)
        for a in addrs[32 * g:32 * g + 32]:
            if a is not None:
                banks[(a // 4) % 32].append(a // 4)
        cyc += max([len(v) for v in banks.values()] + [1])   # every lane on a bank is its own operation, same address or not
    return cyc


# ---- coarse dealing ----
ng_of = [(((minC + q) >> 1) + ((minC + q) & 1) + 3) >> 2 for q in range(nC)]
tasks = []
for q in range(nC):
    for gi in range(ng_of[q]):
        tasks.append((q, gi))
FCG = (len(tasks) + 64 * NWM - 1) // (64 * NWM)


def coarse(oo):
    oD, r = oo // skip, oo % skip
    plr = off_pl + r * plStrideB
    total = ideal = 0
    per_wave_atomic = 0
    for w in range(NWM):
        for g in range(FCG):
            A, B, L = [], [], []
            for lane in range(64):
                T = g * 64 * NWM + ((lane * NWM + w) if DEAL == "spread" else (w * 64 + lane))
                if T < len(tasks):
                    q, gi = tasks[T]
                    A.append(pair_addr(plr, dPl, oD + 8 * gi)); B.append(pair_addr(plr, dPl, oD + minC + q + 8 * gi)); L.append(off_sumC + 4 * q)
                else:
                    A.append(pair_addr(plr, dPl, oD)); B.append(pair_addr(plr, dPl, oD)); L.append(off_sumC)
            for k in range(4):
                total += read_cycles([a + 4 * k for a in A]) + read_cycles([b + 4 * k for b in B]); ideal += 4
            per_wave_atomic += atomic_cycles(L)
    return total, ideal, per_wave_atomic


# ---- refine dealing ----
NLAG = 8 * skip + 1
NCH = (64 * NWM) // NLAG


def refine(oo, lo, hi):
    c0, par, nl = lo >> 1, lo & 1, hi - lo + 1
    G, rho = c0 >> 2, c0 & 3
    NGL = G // NCH
    LG = G - NCH * NGL
    tot = ideal = atom = 0
    for w in range(NWM):
        ap, bp, apx, bpx, app, bpp, sums = [], [], [], [], [], [], []
        for lane in range(64):
            sid = 64 * w + lane
            myT, myC = sid % NLAG, sid // NLAG
            ea = oo + 8 * myC * NGL
            a0 = pair_addr(off_mono, dA, ea); b0 = pair_addr(off_mono, dA, ea + lo + myT)
            xOff = 4 * (NCH * NGL + myC - myC * NGL) * 4
            pOff = (4 * G + myC - 4 * myC * NGL) * 4
            ap.append(a0); bp.append(b0)
            apx.append(a0 + xOff); bpx.append(b0 + xOff if myC < LG else a0 + xOff)
            app.append(a0 + pOff); bpp.append(b0 + pOff if myC < rho else a0 + pOff)
            sums.append(off_sumR + 4 * myT if (myC < NCH and myT < nl) else off_sumR + 4 * myT)
        ng = min(NGL, 3)
        for g in range(ng):
            for k in range(4):
                tot += read_cycles([a + 16 * g + 4 * k for a in ap]) + read_cycles([b + 16 * g + 4 * k for b in bp]); ideal += 4
        for k in range(4):
            tot += read_cycles([a + 4 * k for a in apx]) + read_cycles([b + 4 * k for b in bpx]); ideal += 4
        tot += read_cycles(app) + read_cycles(bpp); ideal += 4
        atom += atomic_cycles(sums)
    return tot, ideal, atom


import random
random.seed(1)
cs = ci = ca = rs = ri = ra = 0
N = 400
for _ in range(N):
    oo = random.randrange(0, wcap - 600)
    bc = random.randrange(4, nC - 2)
    period = (minC + bc) * skip
    lo, hi = max(minP, period - 16), min(maxP, period + 16)
    t, i, a = coarse(oo); cs += t; ci += i; ca += a
    t, i, a = refine(oo, lo, hi); rs += t; ri += i; ra += a
print("pad_mono %d pad_pl %d wcap %d: LDS cycles per step (model, reads only; ragged tasks and sum reads not modelled)" % (pad_mono, pad_pl, wcap))
print("  coarse loads   %6.1f (conflict-free %5.1f)   coarse atomics %5.1f (conflict-free 8)" % (cs / N, ci / N, ca / N))
print("  refine rect    %6.1f (conflict-free %5.1f)   refine atomics %5.1f (conflict-free 8)" % (rs / N, ri / N, ra / N))


# ---- ragged refine tasks (round 4): their two ds_add_u32 per lane ----
def ragged_tasks(par, order):
    tasks = []
    nl = 8 * skip + 1
    if order == "t":          # lag-major (round 3): the tasks of one lag on neighbouring lanes
        for t in range(nl):
            full = (t + par) >> 1
            for r in range(full + ((t + par) & 1)):
                tasks.append(t)
    else:                     # pair-major: neighbouring lanes hold different lags
        rmax = (nl + par) // 2 + 1
        for r in range(rmax):
            for t in range(nl):
                full = (t + par) >> 1
                if r < full + ((t + par) & 1):
                    tasks.append(t)
    return tasks


def ragged_atomics(order, idle):
    tot = 0
    for par in (0, 1):
        tasks = ragged_tasks(par, order)
        rounds = (len(tasks) + 64 * NWM - 1) // (64 * NWM)
        for k in range(rounds):
            for w in range(NWM):
                L = []
                for lane in range(64):
                    T = k * 64 * NWM + 64 * w + lane
                    if T < len(tasks):
                        L.append(off_sumR + 4 * tasks[T])
                    else:
                        L.append(off_sumR if idle == "lag0" else off_sumR + 512 + 4 * lane)
                tot += atomic_cycles(L)
    return tot / 2.0, len(ragged_tasks(0, order)), len(ragged_tasks(1, order))


for order in ("t", "r"):
    for idle in ("lag0", "own"):
        c, n0, n1 = ragged_atomics(order, idle)
        print("  ragged atomics, tasks in %s-major order, idle lanes add to %s: %5.1f LDS cycles per step (%d / %d tasks)" % (
            order, "sums[0]" if idle == "lag0" else "a word of their own", c, n0, n1))
