// Translation unit of the XCD-pipelined decode launches (kernels_xpipe.hip.h: contexts up to 256 keys): 5 block formats x 5 context variants (<= 64 / 128 / 192 / 256 keys, and 257 .. 512 with two workgroups per head) = 25
// persistent kernels here, the 20 resident ones in xpipe_res_tu.hip, the 20 long-context ones (kernels_xlong.hip.h) in four units of five (xlong_*_tu.hip), compiled apart
// from engine.hip so that all of them build in parallel.  The kernel headers define non-inline __global__ functions, so this unit sees them under its own namespace name; the
// parameter block crosses the boundary as bytes (same header, same layout; the size is checked).
#define bgk bgk_xp
#define XP_PART 0
#include "xpipe_tu.inc"
