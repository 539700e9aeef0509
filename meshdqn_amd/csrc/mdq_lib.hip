// Single translation unit of libmeshdqn_hip.so (keeps __constant__ tables and
// the thread-local error string in one place; no relocatable device code needed).
#include "mdq_ipcs.hip"
#include "mdq_pressure_factor.hip"
#include "mdq_gcn.hip"
#include "mdq_gcn_train.hip"
#include "mdq_replay.hip"
#include "mdq_mesh.hip"
#include "mdq_smooth_big.hip"
#include "mdq_smooth.hip"
#include "mdq_smooth_linear.hip"
#include "mdq_topology.hip"
#include "mdq_remesh.hip"
#include "mdq_host_mesh.hip"
