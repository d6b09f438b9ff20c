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
  void* aux;        // op-specific device pointer shared by the clips of a launch (the paired read's unit queue), or nullptr
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
  op.aux = nullptr;
  r->ops.push_back(op);
}

// Copy of an argument block that lives in device memory at a wave-uniform address, read through the
// CONSTANT address space: the loads are invariant scalar loads, the fields land in SGPRs (and can be
// re-loaded where they are used), like kernel arguments.  Read through a generic reference the
// compiler can prove neither the address uniform nor the memory unclobbered and keeps every field in
// VGPRs -- the 64x64 GEMM needed 165 registers instead of 65 (2 waves per SIMD instead of 5).  The
// block must not be written while the launch runs (rmem_launch_recorded's contract).
template <class T>
__device__ __forceinline__ T uniform_copy(const T* src) {
  static_assert(sizeof(T) % 4 == 0, "argument blocks are dword multiples");
  constexpr int NW = sizeof(T) / 4;
  typedef const unsigned __attribute__((address_space(4))) * ConstU;
  const unsigned long u = reinterpret_cast<unsigned long>(src);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  ConstU p = (ConstU)(((unsigned long)hi << 32) | lo);
  unsigned w[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) w[i] = p[i];
  T dst;
  __builtin_memcpy(&dst, w, sizeof(T));
  return dst;
}

template <class A, void (*Body)(const A&, int), int LB>
__global__ __launch_bounds__(LB) void k_one(A a) {
  Body(a, blockIdx.z);
}

template <class A, void (*Body)(const A&, int), int LB>
__global__ __launch_bounds__(LB) void k_many(const char* __restrict__ argv, long stride, int nz) {
  const int clip = blockIdx.z / nz;
  const A& src = *reinterpret_cast<const A*>(argv + (long)clip * stride);
  if constexpr (sizeof(A) <= 640) {
    const A a = uniform_copy(&src);            // fields in SGPRs, like kernel arguments
    Body(a, blockIdx.z - clip * nz);
  } else {
    Body(src, blockIdx.z - clip * nz);         // (large blocks: the body copies what it selects)
  }
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

// BodyMany: the body of the several-clips launch when it differs (argument blocks too large to copy
// whole: the body then copies what it selects with uniform_copy, which must not be applied to a
// kernel argument -- taking its address would move it to scratch).
template <class A, void (*Body)(const A&, int), int LB, void (*BodyMany)(const A&, int) = Body>
int launch(const A& a, dim3 grid, dim3 block, unsigned lds, hipStream_t s) {
  if (Recorder* r = current_recorder()) {
    rec_push(r, &many_thunk<A, BodyMany, LB>, grid, block, lds, &a, (unsigned)sizeof(A));
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
