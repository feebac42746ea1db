// libmdbg.hip — main translation unit of libmdbg_hip.so (gfx950); edges.hip (rocPRIM sort / scans) is compiled separately.
#include "sketch.hip"
#include "table.hip"
#include "synth.hip"
#include "route.hip"
#include "api.inc"
#include "dist_api.inc"
