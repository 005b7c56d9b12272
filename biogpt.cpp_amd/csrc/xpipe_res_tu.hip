// Translation unit of the RESIDENT pipelined decode launches (kernels_xpipe.hip.h with RES = true: biogpt_hip_eval's launch that stays on the
// device between calls): 5 block formats x 5 context variants (<= 64 / 128 / 192 / 256 keys, 257 .. 512 with two workgroups per head).  Same arrangement as xpipe_tu.hip (own namespace name, parameter block as bytes).
#define bgk bgk_xr
#define XP_PART 0
#define XP_LAUNCH bg_xpipe_launch_resident
#define XP_SET_LDS bg_xpipe_set_lds_resident
#include "xpipe_res_tu.inc"
