#define bgk bgk_xl_res_32
#define XL_RES 1
#define XL_KR 32
#define XL_TAG res_32
#include "xlong_tu.inc"
