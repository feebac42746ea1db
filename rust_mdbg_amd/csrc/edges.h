// edges.h — interface between the C ABI (api.inc) and the edge-construction translation unit (edges.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Device memory of the library comes from a process-wide cache of blocks (api.inc): hipMalloc after large frees takes SECONDS on this stack
// (the driver releases memory lazily and the next allocation waits for it: profiles/r04_c_alloc_trace.txt), so freed blocks of 1 MB and more
// are kept and handed out again.  *cap <- usable size (>= bytes).
hipError_t mdbg_block_alloc(void** p, size_t bytes, size_t* cap);
void mdbg_block_free(void* p, size_t cap);

struct EdgeNodes {                 // device-resident node table of the last finalize (rows in index order)
    const uint64_t* keys; const uint32_t* index; const uint16_t* abund; const uint32_t* seqlen; const uint16_t* shift;
    uint64_t n; uint32_t k;
};
struct EdgeBuffers;                // scratch + results, owned by the context (opaque here)
EdgeBuffers* edge_buffers_create();
void edge_buffers_destroy(EdgeBuffers*);

struct EdgeResult {                // device pointers into EdgeBuffers, valid until the next call
    uint64_t n; const uint32_t* n1; const uint8_t* o1; const uint32_t* n2; const uint8_t* o2; const uint32_t* overlap;
    uint64_t presimp_removed;
};
// returns hipSuccess or the failing HIP error; synchronises the stream before returning
hipError_t build_edges(EdgeBuffers* B, const EdgeNodes& nd, float presimp, hipStream_t s, EdgeResult* out);

// sorts every segment [offsets[i], offsets[i+1]) of keys_in ascending into keys_out (rocPRIM segmented radix sort); B provides the scratch
hipError_t sort_segments_u64(EdgeBuffers* B, const uint64_t* keys_in, uint64_t* keys_out, uint64_t n, uint32_t n_segments, const uint32_t* offsets, hipStream_t s);
