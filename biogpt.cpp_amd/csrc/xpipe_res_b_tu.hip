// The second half of xpipe_res_tu.hip: the resident pipelined decode launches of Q5_1 and Q8_0 (xpipe_res_tu.inc).
#define bgk bgk_xr2
#define XP_PART 1
#define XP_LAUNCH bg_xpipe_launch_resident
#define XP_SET_LDS bg_xpipe_set_lds_resident
#include "xpipe_res_tu.inc"
