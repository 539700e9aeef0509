// Translation unit: the three smoothing kernels (mdq_smooth / mdq_smooth_fast hand meshes to each other's kernels).
#include "mdq_smooth_big.hip"
#include "mdq_smooth.hip"
#include "mdq_smooth_linear.hip"
