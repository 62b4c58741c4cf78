/* forwarding header: the reference path "util/common-utils.h" resolves to the engine through include/aslp_compat_kaldi.h (B4 source-level drop-in) */
#include "aslp_compat_kaldi.h"
