// type-only stand-in, see ../vo_cv_shim.h
#include "../vo_cv_shim.h"
