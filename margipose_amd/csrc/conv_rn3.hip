// conv_igemm_k's 96-channel tiles (RN = 3): conv.hip compiled for that part alone (see CV_PART there).
#define CV_PART 3
#include "conv.hip"
