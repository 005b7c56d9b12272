#define bgk bgk_xl_res_64
#define XL_RES 1
#define XL_KR 64
#define XL_TAG res_64
#include "xlong_tu.inc"
