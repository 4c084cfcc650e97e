// bvq_common.hip -- error reporting, tiling and library-level entry points.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "bvq_common.h"

#include <math.h>
#include <string.h>

namespace bvq {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return BVQ_ERR_LAUNCH;
  }
  return BVQ_OK;
}

int default_piece_chunks() {
  // tunable for experiments only; the shipped default is what bench.py measures
  static int v = [] {
    const char* e = getenv("BVQ_PIECE_CHUNKS");
    int n = e ? atoi(e) : 0;
    return (n >= 1 && n <= 4096) ? n : 8;
  }();
  return v;
}

// the quantizer kernels (a store stream next to the loads) on long rows of 4-byte elements: 4 KiB of each stream
// per wave -- the no-arithmetic copy of tools/yardstick.py peaks there too.  The 16-bit kernels keep 8 KiB: with the
// scale-gradient sums they are close to the VALU limit and the per-unit work (scale, reciprocal, two wave reductions)
// of twice as many units costs more than the shorter units give (profiles/r02_per_tensor_pieces.txt).
static int quant_piece_chunks(int vec, int64_t row_len) {
  static int v = [] {
    const char* e = getenv("BVQ_QUANT_PIECE_CHUNKS");  // experiments only
    const int n = e ? atoi(e) : 0;
    return (n >= 1 && n <= 4096) ? n : 0;
  }();
  if (v) return v;
  if (vec == 4) return 4;  // (vec 4 = float32 in 16-byte chunks)
  // 16-bit types: 8 KiB -- except for very long rows (a per-tensor activation), where 7 KiB pieces stream 2-3 %
  // faster than pieces of a power of two (profiles/r02_per_tensor_pieces.txt: 495 -> 510 Gelem/s on the per-tensor
  // headline step); a row of a few pieces (an [8192,8192] weight: two of 8 KiB) stays evenly cut
  // (rows of a few pieces -- weights with a fan-in of 4096..65536: 4 KiB pieces, [8192,8192] bf16 backward 85 -> 81 us,
  //  [4096,11008] 58.2 -> 57.2; 2 KiB and 16 KiB pieces lose 15 %: profiles/r03_weight_pieces.txt)
  const int64_t quantum = (int64_t)kWave * vec;
  if (row_len < 16 * (int64_t)default_piece_chunks() * quantum) return 4;
  return row_len >= 64 * (int64_t)default_piece_chunks() * quantum ? 7 : default_piece_chunks();
}
static int max_units_per_channel_quant() {
  static int v = [] {
    const char* e = getenv("BVQ_QUANT_MAX_UNITS_PER_CHANNEL");  // experiments only
    const int n = e ? atoi(e) : 0;
    return (n >= 1 && n <= (1 << 24)) ? n : (1 << 20);
  }();
  return v;
}

int pick_vec(int max_vec, int64_t rows, int64_t row_len, const void* const* ptrs, const int* elsizes,
             int nptr, bool ragged_ok) {
  // ragged_ok: the caller's kernels walk the (< vec) elements after the last full chunk of EVERY row and
  // tolerate vector accesses that are only element-aligned (gfx9+ under ROCm serves unaligned global
  // accesses in hardware): rows that are not whole chunks, e.g. 14x14 or 7x7 feature maps in 16-bit types,
  // still move 16 bytes per lane
  if (ragged_ok && rows > 1 && row_len >= max_vec && row_len % max_vec != 0) return max_vec;
  int vec = max_vec;
  while (vec > 1) {
    bool ok = (rows == 1) || (row_len % vec == 0);
    for (int i = 0; ok && i < nptr; ++i) {
      if (!ptrs[i]) continue;
      int need = vec * elsizes[i];
      if (need > 16) need = 16;
      if (reinterpret_cast<uintptr_t>(ptrs[i]) % need != 0) ok = false;
    }
    if (ok) break;
    vec >>= 1;
  }
  return vec;
}

int64_t nt_threshold_bytes() {
  static int64_t v = [] {
    const char* e = getenv("BVQ_NT_BYTES");  // experiments only
    const long long n = e ? atoll(e) : -1;
    return n >= 0 ? (int64_t)n : (int64_t)256 << 20;  // the Infinity Cache size
  }();
  return v;
}

// host-side float -> dtype -> float rounding (python scalars that torch converts to the tensor dtype)
float round_host(float f, int dt) {
  if (dt == BVQ_F32 || f != f) return f;
  uint32_t u;
  memcpy(&u, &f, 4);
  if (dt == BVQ_BF16) {
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
  }
  const float a = fabsf(f);
  if (a == 0.f) return f;
  if (a >= 65520.f) return copysignf(INFINITY, f);
  int e;
  frexpf(a, &e);
  int qexp = e - 11;
  if (qexp < -24) qexp = -24;
  const float q = ldexpf(1.f, qexp);
  return copysignf(nearbyintf(a / q) * q, f);
}


// integer environment knob (experiments / kill switches), read by callers once
int env_flag(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

ColsPlan cols_plan(int dtype, int64_t outer, int64_t channels, int64_t inner, bool no_partials, bool team, int vec16) {
  static const int enabled = env_flag("BVQ_COLS", 1);
  ColsPlan p = {};
  const int el = dtype == BVQ_F32 ? 4 : 2;
  // (the backward's workgroup units hold kColsTeamVec16 columns of a 16-bit type per lane: bvq_common.h)
  const int vec = el == 2 ? (vec16 > 0 ? vec16 : (team ? kColsTeamVec16 : 8)) : 4;
  // short inner runs: from 256 bytes per channel row on, the row-mapped units stream well -- unless the rows are
  // not 16-byte multiples (14x14 maps of a 16-bit type: 392 bytes), where the row-mapped route drops to 8- or
  // 2-byte accesses: [1024,1024,14,14] bf16 forward 3.0 -> 5.5 TB/s, abs-max 3.3 -> 5.1 on this route
  // (profiles/r02_column_mapped.txt)
  const int64_t row_bytes = inner * el;
  const bool short_rows = row_bytes < 256;
  const bool ragged_rows = row_bytes % 16 != 0 && row_bytes < 4096 && outer >= 64;
  if (!enabled || channels < 2 || inner < 1 || outer < 2 || !(short_rows || ragged_rows)) return p;
  const int64_t L = channels * inner;
  if (L % (16 / el) != 0 || L / vec > (1 << 30)) return p;  // (the same layouts for every kernel: rows of 16-byte chunks)
  p.rows = outer;
  p.L = L;
  p.vec = vec;
  p.cps = (int32_t)(L / vec);
  p.lpr = p.cps < kWave ? p.cps : kWave;
  p.rpp = kWave / p.lpr;
  p.strips = (p.cps + kWave - 1) / kWave;
  // rows per block: at least 16 chunks per lane, and no more than ~8192 units in all -- every unit leaves a
  // partial row of L entries behind, and those should stay a few percent of the traffic
  int64_t rb = 16 * (int64_t)p.rpp;
  static const int env_units = [] {
    const char* e = getenv("BVQ_COLS_UNITS");  // experiments only
    const int n = e ? atoi(e) : 0;
    return (n >= 64 && n <= (1 << 22)) ? n : 0;
  }();
  // (a kernel that leaves no partial rows behind -- the forward -- is faster with 8 x the units: 4.99 -> 5.70 TB/s
  //  on [802816,512] bf16; with partial rows the extra traffic and the longer fold eat the gain)
  // (team units are workgroups of four waves: the same 65536 waves without partials, 32768 waves with them -- a
  //  workgroup leaves ONE partial row behind, so four times the waves cost no more partial traffic)
  // (read-only kernels with partial rows -- abs-max, min/max, moments: 4096 waves, one resident round of four per SIMD,
  //  measured 7-13 % faster than 8192 on every layout; 2048 and fewer starve the memory system)
  const int want_units = env_units ? env_units : (no_partials ? (team ? 16384 : 65536) : (team ? 8192 : 4096));
  const int64_t want_blocks = want_units / p.strips > 0 ? want_units / p.strips : 1;
  const int64_t rows_for_that = (outer + want_blocks - 1) / want_blocks;
  if (rows_for_that > rb) rb = ((rows_for_that + p.rpp - 1) / p.rpp) * p.rpp;
  if (no_partials && !team && !env_units) {
    // the forward (no partial rows): short blocks by row count -- the resident waves' window of memory again -- and
    // not a power of two (8 / 12 / 16 / 24 rows: [802816,512] float32 0.583 / 0.569 / 0.615 / 0.605 ms, [65536,4096]
    // bf16 0.183 / 0.180 / 0.196 / 0.195; it was 25..98 rows: 0.617 and 0.200 -- profiles/r03_column_mapped.txt)
    static const int env_frows = env_flag("BVQ_COLS_FWD_ROWS", 0);  // experiments only
    rb = (env_frows > 0 ? env_frows : 12) * (int64_t)p.rpp;
  }
  if (team && !no_partials && !env_units) {
    // a workgroup's rows by count, not by a unit total: short blocks keep the resident workgroups' window of memory
    // small, which is what these kernels' bandwidth follows (profiles/r03_column_mapped.txt) -- down to where a wave's
    // set-up (its columns' scales, reciprocals) stops being hidden: float32 (4 columns per lane, seven waves per SIMD)
    // is best at 16 rows; bf16 at 48 and float16 -- whose arithmetic is half again as long -- at 96, both with 4
    // columns per lane and eight waves per SIMD (with 8 columns per lane in ~104 registers, four waves per SIMD, both
    // wanted 96 and streamed 3-8 % slower).
    // One partial row of 4 L bytes per block: 1/48 (1/16) of a block's 3 x 2 L (3 x 4 L) bytes per row = 1.4 % (2 %).
    static const int env_rows = env_flag("BVQ_COLS_TEAM_ROWS", 0);  // experiments only
    const int64_t groups = env_rows > 0 ? env_rows : (dtype == BVQ_BF16 ? 48 : dtype == BVQ_F16 ? 96 : 16);
    rb = groups * (int64_t)p.rpp;
  }
  // a lane's row counter within a unit fits 16 bits (the backward packs it next to a 16-bit key)
  if (rb > 65000 * (int64_t)p.rpp) rb = 65000 * (int64_t)p.rpp;
  // a unit's byte extent fits the 32-bit offsets of a buffer descriptor (the backward addresses it that way)
  const int64_t max_rows = kMaxUnitBytes / (L * 4);
  if (max_rows < p.rpp) return p;
  if (rb > max_rows) rb = (max_rows / p.rpp) * p.rpp;
  p.rb = (int32_t)rb;
  p.nrb = (outer + rb - 1) / rb;
  p.prows = p.nrb * p.rpp;
  p.units = p.nrb * p.strips;
  p.ok = true;
  return p;
}

int max_units_per_channel() {
  static int v = [] {
    const char* e = getenv("BVQ_MAX_UNITS_PER_CHANNEL");
    int n = e ? atoi(e) : 0;
    return (n >= 1 && n <= (1 << 24)) ? n : (1 << 17);
  }();
  return v;
}

static int max_rows_per_unit() {
  static int v = [] {
    const char* e = getenv("BVQ_MAX_RPU");  // experiments only
    const int n = e ? atoi(e) : 0;
    return (n >= 1 && n <= 64) ? n : 64;
  }();
  return v;
}

Tiling make_tiling(int64_t outer, int32_t channels, int64_t row_len, int vec, int64_t unit_cap,
                   bool few_rows) {
  Tiling t;
  t.outer = outer;
  t.channels = channels;
  t.row_len = row_len;
  const int64_t quantum = (int64_t)kWave * vec;  // one 16-byte load per lane
  int64_t piece = (int64_t)default_piece_chunks() * quantum;
  t.rpu = 1;
  t.reverse = 0;
  if (row_len >= piece) {
    if (few_rows) piece = (int64_t)quant_piece_chunks(vec, row_len) * quantum;
    // long rows: cut them into default-sized pieces (a per-tensor quantizer is one very long row: ~10^5
    // units, whose partials the finish kernels combine in two stages), bounded by unit_cap per channel.
    if (unit_cap <= 0) unit_cap = few_rows ? max_units_per_channel_quant() : max_units_per_channel();
    int64_t max_ppr = unit_cap / (outer > 0 ? outer : 1);
    if (max_ppr < 1) max_ppr = 1;
    if (row_len > piece * max_ppr) {
      piece = (row_len + max_ppr - 1) / max_ppr;
      piece = ((piece + quantum - 1) / quantum) * quantum;
    } else if (few_rows && row_len < 16 * piece) {
      // a row of a few pieces: cut it evenly (a float32 56x56 map is 3136 elements: 1280 + 1280 + 576 instead of
      // 3 x 1024 + a 64-element unit)
      int64_t n = (row_len + piece / 2) / piece;
      if (n < 1) n = 1;
      piece = (((row_len + n - 1) / n + quantum - 1) / quantum) * quantum;
    }
  } else {
    // short rows: one piece per row, several rows of one channel per unit, within ~8 pieces worth of
    // work.  The read-only kernels take the row count that wastes the fewest lanes of the 64-wide loads
    // (long units amortise the per-wave setup).  The quantizer kernels (few_rows) take the FEWEST rows
    // that keep >= 86 % of the lanes busy: rows of one channel are channels * row_len apart, and with a
    // store stream next to the loads a unit hopping between them streams measurably worse than one
    // contiguous row (-8 % on [256,512,56,56] bf16; profiles/r01_microbench_v3.txt).
    piece = row_len > 0 ? ((row_len + vec - 1) / vec) * vec : vec;
    const int64_t cpr = row_len / vec;  // full chunks per row
    if (cpr > 0 && outer > 1) {
      const int64_t cap_chunks = 8 * (int64_t)default_piece_chunks() * kWave;
      int64_t best = 1;
      double best_eff = 0.0;
      // (read-only kernels: never so many rows per unit that the launch has fewer than kMinUnits units -- a
      //  [32,512,56,56] shard walked 8 rows at a time is 2048 waves on 1024 SIMDs and streamed at 2.2 TB/s)
      constexpr int64_t kMinUnits = 16384;
      for (int64_t r = 1; r <= outer && r <= max_rows_per_unit() && r * cpr <= cap_chunks; ++r) {
        if (!few_rows && r > 1 && ((outer + r - 1) / r) * channels < kMinUnits) break;
        const int64_t loads = (r * cpr + kWave - 1) / kWave;
        const double eff = (double)(r * cpr) / (double)(loads * kWave);
        if (eff > best_eff + 1e-9) {
          best_eff = eff;
          best = r;
        }
#ifndef BVQ_FEW_ROWS_EFF
#define BVQ_FEW_ROWS_EFF 0.86
#endif
        if (few_rows && eff >= BVQ_FEW_ROWS_EFF) break;
      }
      t.rpu = (int32_t)best;
    }
  }
  t.piece_len = piece;
  t.ppr = row_len > 0 ? (row_len + piece - 1) / piece : 0;
  t.nob = (outer + t.rpu - 1) / t.rpu;
  t.units = t.nob * channels * t.ppr;
  return t;
}

bool cap_unit_extent(Tiling& t, int elsize) {
  const int64_t stride_b = (int64_t)t.channels * t.row_len * elsize;
  const int64_t piece_b = t.piece_len * elsize;
  if (piece_b > kMaxUnitBytes) return false;
  if (t.rpu > 1 && (t.rpu - 1) * stride_b + piece_b > kMaxUnitBytes) {
    t.rpu = (int32_t)(1 + (kMaxUnitBytes - piece_b) / stride_b);
    t.nob = (t.outer + t.rpu - 1) / t.rpu;
    t.units = t.nob * t.channels * t.ppr;
  }
  return true;
}

}  // namespace bvq

extern "C" int bvq_abi_version(void) { return BVQ_ABI_VERSION; }
extern "C" const char* bvq_last_error(void) { return bvq::g_err; }
