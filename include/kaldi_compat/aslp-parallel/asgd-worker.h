/* forwarding header: see include/aslp_compat_kaldi_parallel.h */
#include "aslp_compat_kaldi_parallel.h"
