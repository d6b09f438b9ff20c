// batch.hip -- launch recorder: several clips' memory banks served by ONE launch per kernel
// (SURVEY.md section 8f-2).  See launch.h for the mechanism and include/rmem_hip.h for the contract.
#include "../../include/rmem_hip.h"
#include "launch.h"
#include <string.h>

namespace rmem {
static thread_local Recorder* tl_rec = nullptr;
Recorder* current_recorder() { return tl_rec; }
}  // namespace rmem

using rmem::Recorder;

RmemConfig& rmem_config() {
  static RmemConfig cfg;
  return cfg;
}

// name -> field; unknown names and out-of-range values are refused (nothing changes)
extern "C" int rmem_configure(const char* name, int64_t value) {
  if (!name) return RMEM_ERR_INVALID;
  RmemConfig& c = rmem_config();
  const int v = (int)value;
  struct Key { const char* name; int* field; int lo, hi; };
  const Key keys[] = {{"linear_tiles", &c.linear_tiles, 0, 1}, {"stream_var", &c.stream_var, 1, 4},
                      {"dw_rx", &c.dw_rx, 6, 12}, {"dw_v", &c.dw_v, 1, 4}, {"dw_rows", &c.dw_rows, 0, 4},
                      {"dw_grid_order", &c.dw_grid_order, 0, 1}, {"ida_tokens", &c.ida_tokens, 1, 2}, {"ida_unroll", &c.ida_unroll, 8, 32},
                      {"read_var", &c.read_var, 0, 31}};
  for (const Key& k : keys)
    if (strcmp(name, k.name) == 0) {
      if (value < k.lo || value > k.hi) return RMEM_ERR_INVALID;
      *k.field = v;
      return RMEM_OK;
    }
  return RMEM_ERR_INVALID;
}

extern "C" void* rmem_rec_begin(void) {
  if (rmem::tl_rec) return nullptr;          // recordings do not nest
  rmem::tl_rec = new Recorder();
  return rmem::tl_rec;
}

extern "C" int rmem_rec_end(void* h) {
  if (!h || rmem::tl_rec != h) return RMEM_ERR_INVALID;
  rmem::tl_rec = nullptr;
  return RMEM_OK;
}

extern "C" void rmem_rec_free(void* h) {
  Recorder* r = static_cast<Recorder*>(h);
  if (rmem::tl_rec == r) rmem::tl_rec = nullptr;
  delete r;
}

extern "C" int32_t rmem_rec_count(const void* h) { return h ? (int32_t) static_cast<const Recorder*>(h)->ops.size() : -1; }

extern "C" int64_t rmem_rec_size(const void* h) { return h ? (int64_t) static_cast<const Recorder*>(h)->blob.size() : -1; }

extern "C" const void* rmem_rec_data(const void* h) { return h ? static_cast<const Recorder*>(h)->blob.data() : nullptr; }

// FNV-1a over everything of a recording that is NOT an argument value: which kernels, in which
// order, with which grids and argument-block offsets.  Recordings with equal signatures can share a launch.
extern "C" uint64_t rmem_rec_signature(const void* h) {
  if (!h) return 0;
  const Recorder* r = static_cast<const Recorder*>(h);
  uint64_t x = 1469598103934665603ull;
  auto mix = [&x](uint64_t v) {
    for (int i = 0; i < 8; ++i) {
      x ^= (v >> (8 * i)) & 0xff;
      x *= 1099511628211ull;
    }
  };
  for (const rmem::RecOp& op : r->ops) {
    mix((uint64_t) reinterpret_cast<uintptr_t>(op.fn));
    mix(((uint64_t)op.grid.x << 32) | op.grid.y);
    mix(((uint64_t)op.grid.z << 32) | op.block.x);
    mix(((uint64_t)op.lds << 32) | op.off);
    mix(op.size);
  }
  return x;
}

extern "C" int rmem_launch_recorded(const void* h, const void* dev_args, int64_t clip_stride, int32_t B,
                                    void* stream) {
  if (!h || !dev_args || B <= 0 || B > 4096 || rmem::tl_rec) return RMEM_ERR_INVALID;
  const Recorder* r = static_cast<const Recorder*>(h);
  if (B > 1 && (clip_stride < (int64_t)r->blob.size() || (clip_stride % 16) != 0)) return RMEM_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(dev_args) % 16) != 0) return RMEM_ERR_INVALID;
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (const rmem::RecOp& op : r->ops) {
    if ((long)op.grid.z * B > 65535) return RMEM_ERR_INVALID;
    const int rc = op.fn(op, static_cast<const char*>(dev_args), (long)clip_stride, B, s);
    if (rc != RMEM_OK) return rc;
  }
  return RMEM_OK;
}
