// The three kernels of a batch step -- analysis, tension, walk -- as ONE device code object.
//
// Every .hip source is a code object of its own, loaded wherever the runtime puts it; the three kernels run at the same
// time on the same CUs (concurrent mode) and share the instruction caches.  With one object per kernel their relative
// placement -- and with it the conflict misses of the walk kernel's step loop, the latency-critical chain of the whole
// step -- depended on the ORDER of the sources on the link line (round 2: one order ran the walk kernel 26 % slower with
// byte-identical ISA).  In one object the three are laid out back to back in this order, whatever the link order is.
#include "spx_walk_fast.hip"
#include "spx_tension.hip"
#include "spx_analysis.hip"
