// The second half of xpipe_tu.hip: the pipelined decode launches of Q5_1 and Q8_0 (xpipe_tu.inc).
#define bgk bgk_xp2
#define XP_PART 1
#include "xpipe_tu.inc"
