#define bgk bgk_xl_64
#define XL_RES 0
#define XL_KR 64
#define XL_TAG 64
#include "xlong_tu.inc"
