// libmdbg.hip — single translation unit of libmdbg_hip.so (gfx950).
#include "sketch.hip"
#include "table.hip"
#include "synth.hip"
#include "route.hip"
#include "api.inc"
