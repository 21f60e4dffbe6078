// Unity translation unit of libmp2p_hip.so (one hipcc invocation, gfx950 only).
#include "index_build.hip"
#include "nn_query.hip"
#include "pairs.hip"
#include "nn_pt2pl.hip"
#include "gn_solver.hip"
#include "horn.hip"
#include "filter_decimate.hip"
#include "api.hip"
