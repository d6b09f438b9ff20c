// launch.h -- one launch path for "one clip" and "several clips in one launch".
//
// A kernel of the memory path is written as a device function  Body(const Args&, int bz)  (bz =
// the block's z index inside its own problem).  rmem::launch<Args, Body, LB>() either
//   * launches it for one problem (arguments by value in the kernarg segment), or,
//   * while the calling thread is RECORDING (rmem_rec_begin, batch.hip), appends the argument
//     block and the launch geometry to the recorder and launches nothing.
// A recorded sequence is replayed for B clips by rmem_launch_recorded(): every op becomes ONE
// launch whose grid has B times the z extent; block z / nz picks the clip, and the clip's argument
// block is read from device memory at  args + clip * clip_stride + op offset  (uniform address:
// scalar loads).  The clips' argument blocks come from B recordings of the same code path with
// different buffers -- same ops, same geometry (rmem_rec_signature), different pointers.
//
// The reference cannot batch clips at all: its attention asserts batch 1
// (aot_plus/networks/layers/transformer.py:641,1190), so there is one process per clip stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "rmem_common.h"

namespace rmem {

struct RecOp;
typedef int (*ManyFn)(const RecOp&, const char* dev_args, long clip_stride, int B, hipStream_t s);

struct RecOp {
  ManyFn fn;        // launches the op for B clips
  dim3 grid, block; // geometry of ONE clip
  unsigned lds;     // dynamic LDS bytes
  unsigned off;     // offset of the argument block in the recorder's blob
  unsigned size;
};

struct Recorder {
  std::vector<RecOp> ops;
  std::vector<char> blob;
};

Recorder* current_recorder();   // batch.hip; nullptr = not recording

inline void rec_push(Recorder* r, ManyFn fn, dim3 grid, dim3 block, unsigned lds, const void* args, unsigned size) {
  const unsigned off = (unsigned)((r->blob.size() + 15) & ~(size_t)15);
  r->blob.resize(off + size);
  memcpy(r->blob.data() + off, args, size);
  RecOp op;
  op.fn = fn;
  op.grid = grid;
  op.block = block;
  op.lds = lds;
  op.off = off;
  op.size = size;
  r->ops.push_back(op);
}

template <class A, void (*Body)(const A&, int), int LB>
__global__ __launch_bounds__(LB) void k_one(A a) {
  Body(a, blockIdx.z);
}

template <class A, void (*Body)(const A&, int), int LB>
__global__ __launch_bounds__(LB) void k_many(const char* __restrict__ argv, long stride, int nz) {
  const int clip = blockIdx.z / nz;
  const A& a = *reinterpret_cast<const A*>(argv + (long)clip * stride);
  Body(a, blockIdx.z - clip * nz);
}

template <class A, void (*Body)(const A&, int), int LB>
int many_thunk(const RecOp& op, const char* dev_args, long stride, int B, hipStream_t s) {
  dim3 g = op.grid;
  const int nz = g.z;
  g.z *= B;
  if (op.lds > 0)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_many<A, Body, LB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, op.lds);
  hipLaunchKernelGGL((k_many<A, Body, LB>), g, op.block, op.lds, s, dev_args + op.off, stride, nz);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

template <class A, void (*Body)(const A&, int), int LB>
int launch(const A& a, dim3 grid, dim3 block, unsigned lds, hipStream_t s) {
  if (Recorder* r = current_recorder()) {
    rec_push(r, &many_thunk<A, Body, LB>, grid, block, lds, &a, (unsigned)sizeof(A));
    return RMEM_OK;
  }
  // per launch: the attribute belongs to the (device, function) pair; no process-wide "already set" flag
  if (lds > 0)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_one<A, Body, LB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_one<A, Body, LB>), grid, block, lds, s, a);
  RMEM_CHECK_LAUNCH();
  return RMEM_OK;
}

}  // namespace rmem
