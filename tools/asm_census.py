"""Static census of one kernel's ISA: instructions per basic block by unit (VALU / SALU / LDS / VMEM, fp64 among the VALU), with
the loop nesting the compiler's comments give.  Input: the .s file of `hipcc --save-temps` (make resource-usage leaves none; see
tools/README in DESIGN.md 5.1) and a substring of the mangled kernel name.
Usage: python tools/asm_census.py FILE.s NAME_SUBSTRING [MIN_INSTRUCTIONS]
       python tools/asm_census.py FILE.s NAME_SUBSTRING --scratch     (only the blocks that hold scratch instructions, with what
                                                                       the blocks in front of them branch on)"""
import re
import sys


def census(path, sub, nmin=12):
    s = open(path).read().split('\n')
    starts = [i for i, l in enumerate(s) if re.match(r'^_Z\w+:', l) and sub in l]
    if not starts:
        raise SystemExit("no kernel matches " + sub)
    st = starts[0]
    en = [i for i, l in enumerate(s) if i > st and l.strip().startswith('.Lfunc_end')][0]
    f = s[st:en]
    blk = 'entry'
    stats = {blk: dict(n=0, valu=0, salu=0, lds=0, vmem=0, f64=0, hdr='')}
    order = [blk]
    for l in f:
        m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
        if m:
            blk = m.group(1)
            stats[blk] = dict(n=0, valu=0, salu=0, lds=0, vmem=0, f64=0, hdr=m.group(2).strip())
            order.append(blk)
            continue
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            if t.startswith(';') and ('Loop' in t or 'Depth' in t) and stats[blk]['n'] == 0:
                stats[blk]['hdr'] += ' ' + t
            continue
        d = stats[blk]
        d['n'] += 1
        op = t.split()[0]
        if op.startswith('v_'):
            d['valu'] += 1
        elif op.startswith('s_'):
            d['salu'] += 1
        elif op.startswith('ds_'):
            d['lds'] += 1
        elif op.split('_')[0] in ('global', 'buffer', 'scratch', 'flat'):
            d['vmem'] += 1
        if '_f64' in op:
            d['f64'] += 1
    return f, order, stats


def scratch_blocks(f, order, stats):
    blk = 'entry'
    out = {}
    for l in f:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blk = m.group(1)
            continue
        if 'scratch_' in l:
            out.setdefault(blk, []).append(l.strip().split(';')[0].strip())
    return out


if __name__ == "__main__":
    f, order, stats = census(sys.argv[1], sys.argv[2])
    if "--scratch" in sys.argv:
        sb = scratch_blocks(f, order, stats)
        tot = sum(len(v) for v in sb.values())
        print("%s: %d scratch instructions in %d of %d blocks" % (sys.argv[2], tot, len(sb), len(order)))
        for b in order:
            if b in sb:
                print("  %-12s %2d of %4d instructions  %s" % (b, len(sb[b]), stats[b]['n'], stats[b]['hdr'][:100]))
        raise SystemExit(0)
    nmin = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    tot = dict(n=0, valu=0, salu=0, lds=0, vmem=0, f64=0)
    for b in order:
        d = stats[b]
        for k in tot:
            tot[k] += d[k]
        if d['n'] >= nmin:
            print("%-12s n %4d valu %4d (f64 %3d) salu %3d lds %3d vmem %2d  %s" % (b, d['n'], d['valu'], d['f64'], d['salu'], d['lds'], d['vmem'], d['hdr'][:90]))
    print("total", tot)
