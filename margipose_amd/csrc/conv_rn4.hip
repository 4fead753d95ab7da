// conv_igemm_k's 128-channel tiles (RN = 4): conv.hip compiled for that part alone (see CV_PART there).
#define CV_PART 4
#include "conv.hip"
