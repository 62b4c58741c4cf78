/* forwarding header: see include/aslp_compat_kaldi.h */
#include "aslp_compat_kaldi.h"
