#define bgk bgk_xl_32
#define XL_RES 0
#define XL_KR 32
#define XL_TAG 32
#include "xlong_tu.inc"
