/*
 * ggms_oracle.c -- TEST INFRASTRUCTURE ONLY (see ggms_oracle.h).
 *
 * Plain-C restatement of the reference algorithms on the GGMS hot path.
 * Every function cites the reference file:line (relative to
 * /root/reference/samgraph/common/) it follows.  Nothing here is shipped or
 * measured as the product; xgnn_amd/ must never link or load this file.
 */
#include "ggms_oracle.h"

#include <assert.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================= */
/* cuRAND XORWOW -- third party (CUDA 11.7 curand_kernel.h), restated.     */
/* Call sites: cuda_random_states.cu:44, cuda_sampling_khop3.cu:128, ...   */
/* ======================================================================= */
void orc_xorwow_init(orc_xorwow_t *st, uint64_t seed) {
  /* curand_init(seed, subsequence = 0, offset = 0): no skip-ahead */
  uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u;
  uint32_t s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  st->d = 6615241u + t1 + t0;
  st->v[0] = 123456789u + t0;
  st->v[1] = 362436069u ^ t0;
  st->v[2] = 521288629u + t1;
  st->v[3] = 88675123u ^ t1;
  st->v[4] = 5783321u + t0;
}

uint32_t orc_xorwow_next(orc_xorwow_t *st) {
  uint32_t t = st->v[0] ^ (st->v[0] >> 2);
  st->v[0] = st->v[1];
  st->v[1] = st->v[2];
  st->v[2] = st->v[3];
  st->v[3] = st->v[4];
  st->v[4] = (st->v[4] ^ (st->v[4] << 4)) ^ (t ^ (t << 1));
  st->d += 362437u;
  return st->v[4] + st->d;
}

float orc_xorwow_uniform(orc_xorwow_t *st) {
  /* _curand_uniform: x * 2^-32 + 2^-33, in float (cvt.rn.f32.u32 first) */
  uint32_t x = orc_xorwow_next(st);
  const float k = 2.3283064e-10f; /* CURAND_2POW32_INV == 2^-32 exactly */
  volatile float xf = (float)x;   /* round-to-nearest-even conversion   */
  volatile float p = xf * k;      /* exact (power of two)                */
  return p + (k / 2.0f);
}

double orc_xorwow_uniform_double(orc_xorwow_t *st) {
  /* curand_uniform_double(XORWOW): two draws, _curand_uniform_double_hq */
  uint32_t x = orc_xorwow_next(st);
  uint32_t y = orc_xorwow_next(st);
  uint64_t z = (uint64_t)x ^ ((uint64_t)y << (53 - 32));
  const double k = 1.1102230246251565e-16; /* CURAND_2POW53_INV_DOUBLE */
  return (double)z * k + (k / 2.0);
}

/* batch forms for the long-stream checks of tests/test_xorwow_pin.py */
void orc_xorwow_draws(orc_xorwow_t *st, size_t n, uint32_t *out) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_xorwow_next(st);
}
uint32_t orc_xorwow_fold(orc_xorwow_t *st, size_t n) {
  uint32_t acc = 0;
  for (size_t i = 0; i < n; ++i) acc = (acc << 1 | acc >> 31) ^ orc_xorwow_next(st);
  return acc;
}
void orc_xorwow_uniforms(orc_xorwow_t *st, size_t n, float *out) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_xorwow_uniform(st);
}
void orc_xorwow_uniform_doubles(orc_xorwow_t *st, size_t n, double *out) {
  for (size_t i = 0; i < n; ++i) out[i] = orc_xorwow_uniform_double(st);
}

void orc_random_states_init(orc_xorwow_t *states, size_t num, uint64_t seed) {
  for (size_t t = 0; t < num; ++t) orc_xorwow_init(&states[t], seed + t);
}

/* ======================================================================= */
/* libstdc++ (GCC 11) generators and distributions, restated.              */
/* ======================================================================= */
void orc_mt19937_seed(orc_mt19937_t *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t orc_mt19937_next(orc_mt19937_t *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* bits/uniform_int_dist.h (GCC 11): _S_nd<uint64_t>(urng, uint32 range) --
 * Lemire's nearly-divisionless method, used when the URNG range is exactly
 * 2^32-1 (std::mt19937). */
static uint32_t mt_lemire(orc_mt19937_t *g, uint32_t range) {
  uint64_t product = (uint64_t)orc_mt19937_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)orc_mt19937_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

/* uniform_int_distribution<T>::operator()(mt19937&, {lo,hi}) for urange that
 * fits 32 bits.  urngrange = 0xffffffff. */
static uint64_t mt_uniform(orc_mt19937_t *g, uint64_t lo, uint64_t hi) {
  const uint64_t urngrange = 0xffffffffull;
  uint64_t urange = hi - lo;
  uint64_t ret;
  if (urngrange > urange) {
    uint32_t uerange = (uint32_t)(urange + 1);
    ret = mt_lemire(g, uerange);
  } else if (urngrange < urange) {
    /* upscaling branch: recursive; not reachable for 32-bit id ranges */
    uint64_t tmp;
    do {
      const uint64_t uerngrange = urngrange + 1;
      tmp = uerngrange * mt_uniform(g, 0, urange / uerngrange);
      ret = tmp + (uint64_t)orc_mt19937_next(g);
    } while (ret > urange || ret < tmp);
  } else {
    ret = orc_mt19937_next(g);
  }
  return ret + lo;
}

uint32_t orc_mt19937_uniform_u32(orc_mt19937_t *g, uint32_t lo, uint32_t hi) {
  return (uint32_t)mt_uniform(g, lo, hi);
}

/* bits/stl_algo.h (GCC 11) std::shuffle with a URNG of range 2^32-1 */
void orc_mt19937_shuffle_u32(orc_mt19937_t *g, uint32_t *data, size_t n) {
  if (n == 0) return;
  const uint64_t urngrange = 0xffffffffull;
  const uint64_t urange = (uint64_t)n;
  if (urngrange / urange >= urange) {
    size_t i = 1;
    if ((urange % 2) == 0) {
      size_t j = (size_t)mt_uniform(g, 0, 1);
      uint32_t t = data[i]; data[i] = data[j]; data[j] = t;
      ++i;
    }
    while (i != n) {
      const uint64_t swap_range = (uint64_t)i + 1;
      /* __gen_two_uniform_ints(swap_range, swap_range + 1, g) */
      const uint64_t b0 = swap_range, b1 = swap_range + 1;
      uint64_t x = mt_uniform(g, 0, b0 * b1 - 1);
      size_t p0 = (size_t)(x / b1), p1 = (size_t)(x % b1);
      uint32_t t = data[i]; data[i] = data[p0]; data[p0] = t; ++i;
      t = data[i]; data[i] = data[p1]; data[p1] = t; ++i;
    }
    return;
  }
  for (size_t i = 1; i != n; ++i) {
    size_t j = (size_t)mt_uniform(g, 0, (uint64_t)i);
    uint32_t t = data[i]; data[i] = data[j]; data[j] = t;
  }
}

void orc_minstd0_seed(orc_minstd0_t *g, uint64_t seed) {
  /* linear_congruential_engine<uint_fast32_t,16807,0,2147483647>::seed */
  uint64_t s = seed % 2147483647ull;
  g->x = (s == 0) ? 1 : s;
}

uint64_t orc_minstd0_next(orc_minstd0_t *g) {
  g->x = (g->x * 16807ull) % 2147483647ull;
  return g->x;
}

uint64_t orc_minstd0_uniform_u64(orc_minstd0_t *g, uint64_t lo, uint64_t hi) {
  /* bits/uniform_int_dist.h generic path: urngmin = 1, urngmax = 2^31-2 */
  const uint64_t urngmin = 1, urngrange = 2147483646ull - 1ull;
  const uint64_t urange = hi - lo;
  uint64_t ret;
  if (urngrange > urange) {
    const uint64_t uerange = urange + 1;
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    do {
      ret = orc_minstd0_next(g) - urngmin;
    } while (ret >= past);
    ret /= scaling;
  } else if (urngrange < urange) {
    uint64_t tmp;
    do {
      const uint64_t uerngrange = urngrange + 1;
      tmp = uerngrange * orc_minstd0_uniform_u64(g, 0, urange / uerngrange);
      ret = tmp + (orc_minstd0_next(g) - urngmin);
    } while (ret > urange || ret < tmp);
  } else {
    ret = orc_minstd0_next(g) - urngmin;
  }
  return ret + lo;
}

/* ======================================================================= */
/* Shufflers                                                               */
/* ======================================================================= */
/* dist_shuffler_aligned.cc:89-113 (seed = epoch), cpu_shuffler.cc:68-90,
 * cuda_shuffler.cc (seed = wall clock) */
void orc_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed) {
  orc_minstd0_t g;
  orc_minstd0_seed(&g, seed);
  if (n == 0) return;
  for (size_t i = 0; i < n - 1; ++i) {
    size_t c = (size_t)orc_minstd0_uniform_u64(&g, i, n - 1);
    uint32_t t = data[i]; data[i] = data[c]; data[c] = t;
  }
}

/* dist_shuffler_aligned.cc:46-56 */
size_t orc_aligned_pad(const uint32_t *train, size_t n, size_t num_worker,
                       uint32_t *out) {
  size_t num_data = (n + num_worker - 1) / num_worker * num_worker;
  if (out) {
    memcpy(out, train, n * sizeof(uint32_t));
    for (size_t i = 0; i < num_data - n; ++i) out[i + n] = train[i];
  }
  return num_data;
}

/* dist_shuffler_aligned.cc:123-146 */
void orc_aligned_batch_range(size_t num_local_data, size_t batch_size,
                             size_t epoch, size_t local_step, size_t *offset,
                             size_t *size) {
  size_t off = local_step * batch_size;
  size_t sz = (off + batch_size > num_local_data) ? (num_local_data - off) : batch_size;
  if (epoch == 0 && local_step == 0) {
    sz = (size_t)((double)sz * 1.25);
    sz = (off + sz > num_local_data) ? (num_local_data - off) : sz;
  }
  *offset = off;
  *size = sz;
}

/* ======================================================================= */
/* CPU engine leaves                                                       */
/* ======================================================================= */
/* cpu_random.cc:26-30: static thread_local std::mt19937 (default seed 5489) */
static _Thread_local orc_mt19937_t tl_gen;
static _Thread_local int tl_gen_init = 0;

void orc_cpu_random_reset(void) {
#ifdef _OPENMP
#pragma omp parallel
#endif
  { tl_gen_init = 0; }
  tl_gen_init = 0;
}

uint32_t orc_cpu_random_id(uint32_t lo, uint32_t hi) {
  if (!tl_gen_init) {
    orc_mt19937_seed(&tl_gen, 5489u);
    tl_gen_init = 1;
  }
  return orc_mt19937_uniform_u32(&tl_gen, lo, hi);
}

/* cpu_sampling_khop2.cc:29-76: the CPU twin of the in-place Fisher-Yates sampler.
 * Draw j picks k = RandomID(0, len-j-1) (inclusive bound, so k <= len-j-1),
 * emits indices[off+k], then swaps positions k and len-j-1. */
void orc_cpu_sample_khop2(const orc_id_t *indptr, orc_id_t *indices,
                          const orc_id_t *input, size_t num_input,
                          orc_id_t *out_src, orc_id_t *out_dst,
                          size_t *num_out, size_t fanout, int num_threads) {
  int all_has_fanout = 1;
  (void)num_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads) reduction(&& : all_has_fanout)
#endif
  for (size_t i = 0; i < num_input; ++i) {
    const orc_id_t rid = input[i];
    const orc_id_t off = indptr[rid];
    const orc_id_t len = indptr[rid + 1] - off;
    all_has_fanout = all_has_fanout && (len >= fanout);
    if (len <= fanout) {
      size_t j = 0;
      for (; j < len; ++j) {
        out_src[i * fanout + j] = rid;
        out_dst[i * fanout + j] = indices[off + j];
      }
      for (; j < fanout; ++j) {
        out_src[i * fanout + j] = ORC_EMPTY_KEY;
        out_dst[i * fanout + j] = ORC_EMPTY_KEY;
      }
    } else {
      for (size_t j = 0; j < fanout; ++j) {
        const orc_id_t k = orc_cpu_random_id(0, (orc_id_t)(len - j - 1));
        const orc_id_t picked = indices[off + k];
        out_src[i * fanout + j] = rid;
        out_dst[i * fanout + j] = picked;
        indices[off + k] = indices[off + len - j - 1];
        indices[off + len - j - 1] = picked;
      }
    }
  }
  if (!all_has_fanout) {
    size_t total = num_input * fanout, w = 0;
    for (size_t r = 0; r < total; ++r)
      if (out_src[r] != ORC_EMPTY_KEY) out_src[w++] = out_src[r];
    *num_out = w;
    w = 0;
    for (size_t r = 0; r < total; ++r)
      if (out_dst[r] != ORC_EMPTY_KEY) out_dst[w++] = out_dst[r];
  } else {
    *num_out = num_input * fanout;
  }
}

/* cpu_sampling_khop0.cc:29-83 */
void orc_cpu_sample_khop0(const orc_id_t *indptr, const orc_id_t *indices,
                          const orc_id_t *input, size_t num_input,
                          orc_id_t *out_src, orc_id_t *out_dst,
                          size_t *num_out, size_t fanout, int num_threads) {
  int all_has_fanout = 1;
  (void)num_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads) reduction(&& : all_has_fanout)
#endif
  for (size_t i = 0; i < num_input; ++i) {
    const orc_id_t rid = input[i];
    const orc_id_t off = indptr[rid];
    const orc_id_t len = indptr[rid + 1] - off;
    all_has_fanout = all_has_fanout && (len >= fanout);
    if (len <= fanout) {
      size_t j = 0;
      for (; j < len; ++j) {
        out_src[i * fanout + j] = rid;
        out_dst[i * fanout + j] = indices[off + j];
      }
      for (; j < fanout; ++j) {
        out_src[i * fanout + j] = ORC_EMPTY_KEY;
        out_dst[i * fanout + j] = ORC_EMPTY_KEY;
      }
    } else {
      for (size_t j = 0; j < fanout; ++j) {
        out_src[i * fanout + j] = rid;
        out_dst[i * fanout + j] = indices[off + j];
      }
      for (size_t j = fanout; j < len; ++j) {
        /* RandomID(0, j + 1): inclusive upper bound, as the reference has it */
        const orc_id_t k = orc_cpu_random_id(0, (orc_id_t)(j + 1));
        if (k < fanout) out_dst[i * fanout + k] = indices[off + j];
      }
    }
  }
  if (!all_has_fanout) {
    /* std::remove_if on both arrays, independently (:72-79) */
    size_t total = num_input * fanout, w = 0;
    for (size_t r = 0; r < total; ++r)
      if (out_src[r] != ORC_EMPTY_KEY) out_src[w++] = out_src[r];
    *num_out = w;
    w = 0;
    for (size_t r = 0; r < total; ++r)
      if (out_dst[r] != ORC_EMPTY_KEY) out_dst[w++] = out_dst[r];
  } else {
    *num_out = num_input * fanout;
  }
}

/* cpu_extraction.cc:31-46 / cuda_extraction.cu:31-49: out[i,:] = src[idx[i],:] */
void orc_extract(void *dst, const void *src, const orc_id_t *index,
                 size_t num_index, size_t row_bytes, int num_threads) {
  (void)num_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(num_threads)
#endif
  for (size_t i = 0; i < num_index; ++i) {
    memcpy((char *)dst + i * row_bytes,
           (const char *)src + (size_t)index[i] * row_bytes, row_bytes);
  }
}

/* ======================================================================= */
/* GPU-engine samplers                                                     */
/* ======================================================================= */
/* common.cc:488-497 */
size_t orc_predict_num_nodes(size_t batch_size, const size_t *fanout,
                             size_t num_fanout_to_comp) {
  size_t count = batch_size;
  for (int i = (int)num_fanout_to_comp - 1; i >= 0; --i) count += count * fanout[i];
  return count;
}

/* count_edge + compact_edge (cuda_sampling_khop3.cu:148-230; identical copies
 * in khop0.cu:157-239): per seed, the leading non-empty slots, seed order. */
static size_t compact_tmp(const orc_id_t *tmp_src, const orc_id_t *tmp_dst,
                          size_t num_input, size_t fanout, orc_id_t *out_src,
                          orc_id_t *out_dst) {
  size_t w = 0;
  for (size_t index = 0; index < num_input; ++index) {
    size_t cnt = 0;
    for (size_t j = 0; j < fanout; ++j)
      if (tmp_src[index * fanout + j] != ORC_EMPTY_KEY) ++cnt;
    for (size_t j = 0; j < cnt; ++j) {
      out_src[w] = tmp_src[index * fanout + j];
      out_dst[w] = tmp_dst[index * fanout + j];
      ++w;
    }
  }
  return w;
}

/* cuda_sampling_khop3.cu:76-146.  Block = 8 groups of 16 lanes; group
 * (b, y) owns RNG state i = 8b + y and walks seeds 128b + y + 8k, k = 0..15,
 * in order.  Canonical lock-step semantics: the 16 lanes draw the same value
 * from the shared state, so one draw == one insert attempt (:127-130). */
void orc_sample_khop3(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  const size_t GROUPS = 8, TILE = 128;
  assert(fanout < 128); /* :85 */
  orc_id_t *tmp_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  orc_id_t *tmp_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  const size_t num_blocks = (num_input + TILE - 1) / TILE;
  uint32_t chosen[128];
  for (size_t b = 0; b < num_blocks; ++b) {
    for (size_t y = 0; y < GROUPS; ++y) {
      const size_t i = b * GROUPS + y;
      assert(i < num_states);
      (void)num_states;
      orc_xorwow_t st = states[i];
      for (size_t index = TILE * b + y; index < TILE * (b + 1); index += GROUPS) {
        if (index >= num_input) continue;
        const orc_id_t rid = input[index];
        const orc_id_t off = indptr[rid];
        const orc_id_t len = indptr[rid + 1] - off;
        if (len <= fanout) {
          size_t j = 0;
          for (; j < len; ++j) {
            tmp_src[index * fanout + j] = rid;
            tmp_dst[index * fanout + j] = indices[off + j];
          }
          for (; j < fanout; ++j) {
            tmp_src[index * fanout + j] = ORC_EMPTY_KEY;
            tmp_dst[index * fanout + j] = ORC_EMPTY_KEY;
          }
        } else {
          size_t count = 0;
          while (count < fanout) {
            uint32_t r = orc_xorwow_next(&st) % len;
            int dup = 0;
            for (size_t c = 0; c < count; ++c)
              if (chosen[c] == r) { dup = 1; break; }
            if (!dup) chosen[count++] = r; /* insertion order == items[] order */
          }
          for (size_t j = 0; j < fanout; ++j) {
            tmp_src[index * fanout + j] = rid;
            tmp_dst[index * fanout + j] = indices[off + chosen[j]];
          }
        }
      }
      states[i] = st;
    }
  }
  *num_out = compact_tmp(tmp_src, tmp_dst, num_input, fanout, out_src, out_dst);
  free(tmp_src);
  free(tmp_dst);
}

/* cuda_sampling_khop2.cu:46-95 (ORIGIN_KHOP2 is #defined at :35).  Block b = 256
 * threads over the tile [1024b, 1024b+1024); thread t owns RNG state i = 256b + t
 * and serves seeds 1024b + t + 256r, r = 0..3, in order.  A seed with more than
 * `fanout` neighbours takes a partial Fisher-Yates pass over ITS OWN slice of
 * `indices`, in place: draw j picks position curand % (len - j), emits it, and
 * swaps it with position len-j-1.  The CSR is therefore mutated, and the next
 * call sees the permuted lists.  Deterministic as long as the seeds of one call
 * are distinct (they are: layer inputs come out of the dedup table). */
void orc_sample_khop2(const orc_id_t *indptr, orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  const size_t BLOCK = 256, TILE = 1024;
  orc_id_t *tmp_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  orc_id_t *tmp_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  const size_t num_blocks = (num_input + TILE - 1) / TILE;
  for (size_t b = 0; b < num_blocks; ++b) {
    for (size_t t = 0; t < BLOCK; ++t) {
      const size_t i = b * BLOCK + t;
      assert(i < num_states); /* :57 */
      (void)num_states;
      orc_xorwow_t st = states[i];
      for (size_t index = TILE * b + t; index < TILE * (b + 1); index += BLOCK) {
        if (index >= num_input) continue;
        const orc_id_t rid = input[index];
        const orc_id_t off = indptr[rid];
        const orc_id_t len = indptr[rid + 1] - off;
        orc_id_t *ts = tmp_src + index * fanout, *td = tmp_dst + index * fanout;
        if (len <= fanout) {
          size_t j = 0;
          for (; j < len; ++j) { ts[j] = rid; td[j] = indices[off + j]; }
          for (; j < fanout; ++j) { ts[j] = ORC_EMPTY_KEY; td[j] = ORC_EMPTY_KEY; }
        } else {
          for (size_t j = 0; j < fanout; ++j) {
            const size_t sel = orc_xorwow_next(&st) % (len - j);
            const orc_id_t picked = indices[off + sel];
            ts[j] = rid;
            td[j] = picked;
            indices[off + sel] = indices[off + len - j - 1];
            indices[off + len - j - 1] = picked;
          }
        }
      }
      states[i] = st;
    }
  }
  *num_out = compact_tmp(tmp_src, tmp_dst, num_input, fanout, out_src, out_dst);
  free(tmp_src);
  free(tmp_dst);
}

/* cuda_sampling_khop0.cu:102-153 (NEW_ALGO is #defined at :37).  Block =
 * 4 warps of 32 lanes, 64 seeds per block; lane (b, x, w) seeds a fresh
 * generator with (b*128 + x*4 + w) + num_input and keeps it across the warp's
 * 16 seeds.  Position j >= fanout of a seed is handled by lane j % 32.
 * Canonical collision rule for the atomicExch at :144-148: highest j wins,
 * i.e. the slots end as if j were processed in ascending order. */
void orc_sample_khop0(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  const size_t WARP = 32, BLOCK_WARP = 4, TILE = 64;
  orc_id_t *tmp_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  orc_id_t *tmp_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  const size_t num_blocks = (num_input + TILE - 1) / TILE;
  orc_xorwow_t lane_state[32];
  for (size_t b = 0; b < num_blocks; ++b) {
    for (size_t w = 0; w < BLOCK_WARP; ++w) {
      for (size_t x = 0; x < WARP; ++x) {
        size_t i = b * WARP * BLOCK_WARP + x * BLOCK_WARP + w;
        orc_xorwow_init(&lane_state[x], (uint64_t)(i + num_input));
      }
      size_t last = TILE * (b + 1) < num_input ? TILE * (b + 1) : num_input;
      for (size_t index = TILE * b + w; index < last; index += BLOCK_WARP) {
        const orc_id_t rid = input[index];
        const orc_id_t off = indptr[rid];
        const orc_id_t len = indptr[rid + 1] - off;
        if (len <= fanout) {
          size_t j = 0;
          for (; j < len; ++j) {
            tmp_src[index * fanout + j] = rid;
            tmp_dst[index * fanout + j] = indices[off + j];
          }
          for (; j < fanout; ++j) {
            tmp_src[index * fanout + j] = ORC_EMPTY_KEY;
            tmp_dst[index * fanout + j] = ORC_EMPTY_KEY;
          }
        } else {
          for (size_t j = 0; j < fanout; ++j) {
            tmp_src[index * fanout + j] = rid;
            tmp_dst[index * fanout + j] = indices[off + j];
          }
          for (size_t j = fanout; j < len; ++j) {
            size_t k = (size_t)orc_xorwow_next(&lane_state[j % WARP]) % (j + 1);
            if (k < fanout) tmp_dst[index * fanout + k] = indices[off + j];
          }
        }
      }
    }
  }
  *num_out = compact_tmp(tmp_src, tmp_dst, num_input, fanout, out_src, out_dst);
  free(tmp_src);
  free(tmp_dst);
}

/* stable LSD radix == any stable sort by 32-bit key; merge sort on (key, pos) */
typedef struct { uint32_t key; uint32_t val; } kv32_t;
static void stable_sort_kv32(kv32_t *a, size_t n) {
  if (n < 2) return;
  kv32_t *tmp = (kv32_t *)malloc(sizeof(kv32_t) * n);
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      size_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, sizeof(kv32_t) * n);
  }
  free(tmp);
}

/* cuda_sampling_weighted_khop.cu:41-76 (sample), :172-181 (stable radix sort
 * by src), :78-128 (adjacent-duplicate compaction), launch :156-164. */
static void sample_with_replacement(const orc_id_t *indptr, const orc_id_t *indices,
                                    const float *prob_table, /* NULL: uniform (khop1) */
                                    const orc_id_t *alias_table, /* NULL with prob_table: prefix sums */
                                    const orc_id_t *input, size_t num_input,
                                    size_t fanout, orc_xorwow_t *states,
                                    size_t num_states, orc_id_t *out_src,
                                    orc_id_t *out_dst, size_t *num_out) {
  const size_t num_task = num_input * fanout;
  const size_t kMaxThreads = 512 * 1024, kBlock = 256; /* constant.h:66,72 */
  if (num_task == 0) { *num_out = 0; return; }
  size_t num_threads = num_task < kMaxThreads ? num_task : kMaxThreads;
  size_t span = (num_threads + kBlock - 1) / kBlock * kBlock; /* grid*block */
  kv32_t *kv = (kv32_t *)malloc(sizeof(kv32_t) * num_task);
  /* tmp_dst of len == 0 tasks is never written by the kernel (:58-60); its
   * value never reaches the output (the src test masks it).  Use EMPTY. */
  for (size_t t = 0; t < num_task; ++t) { kv[t].key = ORC_EMPTY_KEY; kv[t].val = ORC_EMPTY_KEY; }
  for (size_t tid = 0; tid < span && tid < num_task; ++tid) {
    assert(tid < num_states);
    (void)num_states;
    orc_xorwow_t st = states[tid];
    for (size_t task = tid; task < num_task; task += span) {
      const orc_id_t rid = input[task / fanout];
      const orc_id_t off = indptr[rid];
      const orc_id_t len = indptr[rid + 1] - indptr[rid];
      if (len == 0) {
        kv[task].key = ORC_EMPTY_KEY;
      } else {
        kv[task].key = rid;
        if (prob_table && !alias_table) {
          /* cuda_sampling_weighted_khop_prefix.cu:59-88: inverse-CDF draw over the per-list prefix sums */
          const float upbound = prob_table[off + len - 1];
          const float x = orc_xorwow_uniform(&st) * upbound;
          if (x <= prob_table[off]) {
            kv[task].val = indices[off];
          } else {
            size_t lo = off, hi = off + len - 1;
            while (hi - lo >= 2) {
              const size_t mid = (lo + hi) >> 1;
              if (prob_table[mid] >= x) hi = mid; else lo = mid;
            }
            kv[task].val = indices[hi];
          }
        } else {
          size_t k = orc_xorwow_next(&st) % len;
          if (prob_table) {
            float r = orc_xorwow_uniform(&st);
            kv[task].val = (r < prob_table[off + k]) ? indices[off + k] : alias_table[off + k];
          } else {
            kv[task].val = indices[off + k]; /* khop1.cu:65-67 */
          }
        }
      }
    }
    states[tid] = st;
  }
  stable_sort_kv32(kv, num_task);
  size_t w = 0;
  for (size_t t = 0; t < num_task; ++t) {
    int cond;
    if (t < num_task - 1)
      cond = (kv[t].key != kv[t + 1].key || kv[t].val != kv[t + 1].val) && kv[t].key != ORC_EMPTY_KEY;
    else
      cond = kv[t].key != ORC_EMPTY_KEY;
    if (cond) { out_src[w] = kv[t].key; out_dst[w] = kv[t].val; ++w; }
  }
  *num_out = w;
  free(kv);
}

void orc_sample_weighted_khop(const orc_id_t *indptr, const orc_id_t *indices,
                              const float *prob_table,
                              const orc_id_t *alias_table,
                              const orc_id_t *input, size_t num_input,
                              size_t fanout, orc_xorwow_t *states,
                              size_t num_states, orc_id_t *out_src,
                              orc_id_t *out_dst, size_t *num_out) {
  assert(prob_table && alias_table);
  sample_with_replacement(indptr, indices, prob_table, alias_table, input, num_input, fanout, states, num_states,
                          out_src, out_dst, num_out);
}

/* cuda_sampling_weighted_khop_prefix.cu:41-91 (one curand_uniform per task, binary search in the list's
 * prefix-sum table), then the same sort / adjacent-duplicate compaction :93-142,185-246. */
void orc_sample_weighted_khop_prefix(const orc_id_t *indptr, const orc_id_t *indices,
                                     const float *prob_prefix_table,
                                     const orc_id_t *input, size_t num_input, size_t fanout,
                                     orc_xorwow_t *states, size_t num_states,
                                     orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  assert(prob_prefix_table);
  sample_with_replacement(indptr, indices, prob_prefix_table, NULL, input, num_input, fanout, states, num_states,
                          out_src, out_dst, num_out);
}

/* cuda_sampling_weighted_khop_hash_dedup.cu:41-57,59-117: thread t of block b owns state 256 b + t and seeds
 * 1024 b + t + 256 r (the khop2 idiom).  A seed with more than `fanout` neighbours draws alias-method
 * candidates (two draws each) until `fanout` DISTINCT ids are found; the per-thread 50-slot table keys its
 * entries by round id = the seed's id, so it is never cleared (entries of an earlier seed with the same id
 * would still count -- kept as is).  Note the strict `r > prob` here against `r < prob` in the plain sampler. */
void orc_sample_weighted_khop_hash_dedup(const orc_id_t *indptr, const orc_id_t *indices,
                                         const float *prob_table, const orc_id_t *alias_table,
                                         const orc_id_t *input, size_t num_input, size_t fanout,
                                         orc_xorwow_t *states, size_t num_states,
                                         orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  const size_t BLOCK = 256, TILE = 1024, SLOTS = 50;
  assert(fanout < SLOTS);
  orc_id_t *tmp_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  orc_id_t *tmp_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_input * fanout + 1));
  const size_t num_blocks = (num_input + TILE - 1) / TILE;
  for (size_t b = 0; b < num_blocks; ++b) {
    for (size_t t = 0; t < BLOCK; ++t) {
      const size_t i = b * BLOCK + t;
      assert(i < num_states);
      (void)num_states;
      orc_xorwow_t st = states[i];
      orc_id_t val[50], round_of[50];
      for (size_t z = 0; z < SLOTS; ++z) { val[z] = ORC_EMPTY_KEY; round_of[z] = ORC_EMPTY_KEY; }
      for (size_t index = TILE * b + t; index < TILE * (b + 1); index += BLOCK) {
        if (index >= num_input) continue;
        const orc_id_t rid = input[index];
        const orc_id_t off = indptr[rid];
        const orc_id_t len = indptr[rid + 1] - off;
        orc_id_t *ts = tmp_src + index * fanout, *td = tmp_dst + index * fanout;
        if (len <= fanout) {
          size_t j = 0;
          for (; j < len; ++j) { ts[j] = rid; td[j] = indices[off + j]; }
          for (; j < fanout; ++j) { ts[j] = ORC_EMPTY_KEY; td[j] = ORC_EMPTY_KEY; }
          continue;
        }
        size_t got = 0, tries = 0;
        while (got < fanout) {
          /* the reference spins forever on a list with fewer than `fanout` distinct candidates; after
           * ORC_HASH_DEDUP_MAX_TRIES draws for one seed every candidate is taken (same rule in the HIP kernel) */
          const int give_up = ++tries > ORC_HASH_DEDUP_MAX_TRIES;
          const size_t k = orc_xorwow_next(&st) % len;
          const float r = orc_xorwow_uniform(&st);
          orc_id_t cand = indices[off + k];
          if (r > prob_table[off + k]) cand = alias_table[off + k];
          /* insert_hash_table, :41-57 */
          size_t pos = cand % SLOTS, gap = 1;
          int is_new = 1; /* bounded probing: the reference loops forever once the 50 slots are full */
          for (size_t probe = 0; probe < ORC_HASH_DEDUP_MAX_PROBES; ++probe) {
            if (round_of[pos] != rid) { round_of[pos] = rid; val[pos] = cand; break; }
            if (val[pos] == cand) { is_new = 0; break; }
            pos = (pos + gap) % SLOTS;
            ++gap;
          }
          if (!is_new && !give_up) continue;
          ts[got] = rid;
          td[got] = cand;
          ++got;
        }
      }
      states[i] = st;
    }
  }
  *num_out = compact_tmp(tmp_src, tmp_dst, num_input, fanout, out_src, out_dst);
  free(tmp_src);
  free(tmp_dst);
}

/* cuda_sampling_khop1.cu:42-72 (one curand % len per task, with replacement; grid-stride over
 * <= kKHop1MaxThreads = 512 K stored states), SortPairs by src :160-176, count_edge/compact_edge
 * :74-127 (an entry equal to its successor is dropped) -- the weighted sampler minus the alias draw. */
void orc_sample_khop1(const orc_id_t *indptr, const orc_id_t *indices,
                      const orc_id_t *input, size_t num_input, size_t fanout,
                      orc_xorwow_t *states, size_t num_states,
                      orc_id_t *out_src, orc_id_t *out_dst, size_t *num_out) {
  sample_with_replacement(indptr, indices, NULL, NULL, input, num_input, fanout, states, num_states, out_src, out_dst,
                          num_out);
}

/* cuda_sampling_random_walk.cu:43-112; launch geometry :137-141 */
void orc_random_walk_raw(const orc_id_t *indptr, const orc_id_t *indices,
                         const orc_id_t *input, size_t num_input,
                         size_t walk_length, double restart_prob,
                         size_t num_walk, orc_xorwow_t *states,
                         size_t num_states, orc_id_t *tmp_src,
                         orc_id_t *tmp_dst) {
  size_t bx = 256, by = 1;
  while (bx >= 2 * num_walk) { bx /= 2; by *= 2; }
  const size_t grid = (num_input + by - 1) / by;
  for (size_t blk = 0; blk < grid; ++blk) {
    for (size_t tx = 0; tx < bx; ++tx) {
      for (size_t ty = 0; ty < by; ++ty) {
        size_t thread_id = bx * by * blk + by * tx + ty;
        assert(thread_id < num_states);
        (void)num_states;
        orc_xorwow_t st = states[thread_id];
        size_t node_idx = blk * by + ty;
        const size_t stride = by * grid;
        while (node_idx < num_input) {
          orc_id_t start = input[node_idx];
          for (size_t walk = tx; walk < num_walk; walk += bx) {
            orc_id_t node = start;
            for (size_t step = 0; step < walk_length; ++step) {
              size_t pos = node_idx * num_walk * walk_length + step * num_walk + walk;
              if (node == ORC_EMPTY_KEY) {
                tmp_src[pos] = ORC_EMPTY_KEY;
              } else {
                const orc_id_t off = indptr[node];
                const orc_id_t len = indptr[node + 1] - off;
                if (len == 0) {
                  tmp_src[pos] = ORC_EMPTY_KEY;
                  node = ORC_EMPTY_KEY;
                } else {
                  size_t k = orc_xorwow_next(&st) % len;
                  tmp_src[pos] = start;
                  tmp_dst[pos] = indices[off + k];
                  node = indices[off + k];
                  if (orc_xorwow_uniform_double(&st) < restart_prob) node = ORC_EMPTY_KEY;
                }
              }
            }
          }
          node_idx += stride;
        }
        states[thread_id] = st;
      }
    }
  }
}

/* FrequencyHashmap::GetTopK, cuda_frequency_hashmap.cu:643-841.  Per seed
 * position: unique visited nodes with counts; order = count descending, ties
 * by first visit (index) ascending -- the stable descending radix sort of
 * key ((num_seed - seed_idx) << 32 | count) at :733-746 over the list that
 * generate_unique_edges_pos :377-425 emits in ascending winner index.
 * Canonical winner of a duplicate (seed, dst) = first occurrence. */
static void topk(const orc_id_t *tmp_src, const orc_id_t *tmp_dst,
                 const orc_id_t *input, size_t num_input, size_t per_node,
                 size_t K, orc_id_t *out_src, orc_id_t *out_dst,
                 orc_id_t *out_data, size_t *num_out) {
  size_t w = 0;
  orc_id_t *uniq = (orc_id_t *)malloc(sizeof(orc_id_t) * (per_node + 1));
  orc_id_t *cnt = (orc_id_t *)malloc(sizeof(orc_id_t) * (per_node + 1));
  char *taken = (char *)malloc(per_node + 1);
  for (size_t s = 0; s < num_input; ++s) {
    size_t nu = 0;
    for (size_t e = 0; e < per_node; ++e) {
      size_t idx = s * per_node + e;
      if (tmp_src[idx] == ORC_EMPTY_KEY) continue;
      orc_id_t d = tmp_dst[idx];
      size_t u = 0;
      for (; u < nu; ++u) if (uniq[u] == d) break;
      if (u == nu) { uniq[nu] = d; cnt[nu] = 1; ++nu; } else { ++cnt[u]; }
    }
    size_t take = nu < K ? nu : K;
    memset(taken, 0, nu + 1);
    for (size_t k = 0; k < take; ++k) {
      size_t best = (size_t)-1;
      for (size_t u = 0; u < nu; ++u) {
        if (taken[u]) continue;
        if (best == (size_t)-1 || cnt[u] > cnt[best]) best = u; /* ties: first */
      }
      taken[best] = 1;
      out_src[w] = input[s];
      out_dst[w] = uniq[best];
      out_data[w] = cnt[best];
      ++w;
    }
  }
  *num_out = w;
  free(uniq); free(cnt); free(taken);
}

void orc_sample_random_walk(const orc_id_t *indptr, const orc_id_t *indices,
                            const orc_id_t *input, size_t num_input,
                            size_t walk_length, double restart_prob,
                            size_t num_walk, size_t K, orc_xorwow_t *states,
                            size_t num_states, orc_id_t *out_src,
                            orc_id_t *out_dst, orc_id_t *out_data,
                            size_t *num_out) {
  size_t total = num_input * num_walk * walk_length;
  orc_id_t *ts = (orc_id_t *)malloc(sizeof(orc_id_t) * (total + 1));
  orc_id_t *td = (orc_id_t *)malloc(sizeof(orc_id_t) * (total + 1));
  orc_random_walk_raw(indptr, indices, input, num_input, walk_length, restart_prob,
                      num_walk, states, num_states, ts, td);
  topk(ts, td, input, num_input, num_walk * walk_length, K, out_src, out_dst, out_data, num_out);
  free(ts); free(td);
}

/* ======================================================================= */
/* Ordered hash table                                                      */
/* ======================================================================= */
struct orc_hashtable {
  orc_id_t *local_of; /* id -> local, EMPTY if absent (cpu_hashtable2.cc layout) */
  size_t num_node;
  orc_id_t *n2o;
  size_t cap;
  size_t num_items;
};

orc_hashtable_t *orc_ht_create(size_t num_node, size_t capacity) {
  orc_hashtable_t *ht = (orc_hashtable_t *)malloc(sizeof(*ht));
  ht->local_of = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_node + 1));
  memset(ht->local_of, 0xff, sizeof(orc_id_t) * (num_node + 1));
  ht->n2o = (orc_id_t *)malloc(sizeof(orc_id_t) * (capacity + 1));
  ht->num_node = num_node;
  ht->cap = capacity;
  ht->num_items = 0;
  return ht;
}

void orc_ht_destroy(orc_hashtable_t *ht) {
  if (!ht) return;
  free(ht->local_of);
  free(ht->n2o);
  free(ht);
}

/* cuda_hashtable.cu:739-742 (version bump) / cpu_hashtable2.cc:183-191 */
void orc_ht_reset(orc_hashtable_t *ht) {
  for (size_t i = 0; i < ht->num_items; ++i) ht->local_of[ht->n2o[i]] = ORC_EMPTY_KEY;
  ht->num_items = 0;
}

/* cuda_hashtable.cu:744-837 (FillWithDuplicates), :850-912
 * (FillWithDupRevised), cpu_hashtable2.cc:53-107 (Populate): keys not yet in
 * the table get consecutive local ids, ordered by the input index of the
 * winning instance; canonical winner = first occurrence. */
size_t orc_ht_fill_with_duplicates(orc_hashtable_t *ht, const orc_id_t *input,
                                   size_t num_input) {
  for (size_t i = 0; i < num_input; ++i) {
    orc_id_t id = input[i];
    assert(id < ht->num_node);
    if (ht->local_of[id] == ORC_EMPTY_KEY) {
      assert(ht->num_items < ht->cap);
      ht->local_of[id] = (orc_id_t)ht->num_items;
      ht->n2o[ht->num_items++] = id;
    }
  }
  return ht->num_items;
}

size_t orc_ht_num_items(const orc_hashtable_t *ht) { return ht->num_items; }
const orc_id_t *orc_ht_unique(const orc_hashtable_t *ht) { return ht->n2o; }

/* cuda_mapping.cu:49-66 */
void orc_ht_map_edges(const orc_hashtable_t *ht, const orc_id_t *src,
                      const orc_id_t *dst, size_t num_edges, orc_id_t *new_src,
                      orc_id_t *new_dst) {
  for (size_t i = 0; i < num_edges; ++i) {
    new_src[i] = ht->local_of[src[i]];
    new_dst[i] = ht->local_of[dst[i]];
  }
}

/* ----------------------------------------------------------------------- */
/* Weight tables of the weighted samplers (dataset tools).
 * utility/data-process/toolkit/weight/create_alias_table.cc:105-170: per list -- float weights normalised to
 * mean 1 (:131-135), two FIFO work lists (std::queue, :138-147), a small slot takes its own neighbour with
 * probability w and the front large one's NODE ID otherwise, the large one gives up 1 - w (:149-166), leftovers
 * accept with probability 1 (:168-180; their alias slot keeps the vector's initial 0, :212).
 * create_prob_prefix_table.cc:94-123: running float sum per list. */
void orc_create_alias_table(const orc_id_t *indptr, const orc_id_t *indices, size_t num_node, const float *weights,
                            float *prob_table, orc_id_t *alias_table) {
  for (size_t v = 0; v < num_node; ++v) {
    const uint32_t off = indptr[v], len = indptr[v + 1] - off;
    float *w = (float *)malloc(sizeof(float) * (len + 1));
    uint32_t *qs = (uint32_t *)malloc(sizeof(uint32_t) * (2 * (size_t)len + 2)); /* every index enters a queue <= 2 times */
    uint32_t *ql = (uint32_t *)malloc(sizeof(uint32_t) * (2 * (size_t)len + 2));
    size_t sh = 0, st = 0, lh = 0, lt = 0;
    volatile float sum = 0.0f; /* volatile: no reassociation, the tool sums in index order */
    for (uint32_t i = 0; i < len; ++i) sum = sum + weights[off + i];
    for (uint32_t i = 0; i < len; ++i) {
      volatile float x = weights[off + i] / sum;
      x = x * (float)len;
      w[i] = x;
      alias_table[off + i] = 0;
    }
    for (uint32_t i = 0; i < len; ++i) {
      if (w[i] < 1.0f) qs[st++] = i; else ql[lt++] = i;
    }
    while (sh < st && lh < lt) {
      const uint32_t s = qs[sh++], l = ql[lh++];
      prob_table[off + s] = w[s];
      alias_table[off + s] = indices[off + l];
      volatile float give = 1.0f - w[s];
      volatile float rest = w[l] - give;
      w[l] = rest;
      if (w[l] < 1.0f) qs[st++] = l; else ql[lt++] = l;
    }
    while (lh < lt) prob_table[off + ql[lh++]] = 1.0f;
    while (sh < st) prob_table[off + qs[sh++]] = 1.0f;
    free(w); free(qs); free(ql);
  }
}

void orc_create_prob_prefix_table(const orc_id_t *indptr, size_t num_node, const float *weights, float *prefix) {
  for (size_t v = 0; v < num_node; ++v) {
    const uint32_t off = indptr[v], len = indptr[v + 1] - off;
    volatile float sum = 0.0f;
    for (uint32_t i = 0; i < len; ++i) {
      sum = sum + weights[off + i];
      prefix[off + i] = sum;
    }
  }
}

/* ----------------------------------------------------------------------- */
/* CPUHashTable2 with its OpenMP structure (cpu/cpu_hashtable2.cc:35-191): a direct-indexed bucket per node
 * id {key, local, index, version}; Populate = CAS claim, per-thread count, serial prefix over threads, per-thread
 * id assignment (static schedule, so a thread's items are a contiguous index range and ids stay ordered by
 * input index within and across threads); MapEdges / Reset parallel loops.  With one thread this is exactly
 * orc_ht_fill_with_duplicates; with several the winner among duplicates is whichever CAS lands first, as in the
 * reference.  Used by bench.py's cpu_baseline so that the dedup/remap leg runs on as many cores as the
 * sampler and extract legs (the reference's default hash table is this one, run_config.cc:56). */
typedef struct { orc_id_t key, local, index, version; } orc_bucket2_t;
struct orc_cpu_ht2 {
  orc_bucket2_t *o2n;
  orc_id_t *n2o;
  size_t num_node, num_items;
  orc_id_t version;
};

orc_cpu_ht2_t *orc_cpu_ht2_create(size_t num_node, int threads) {
  orc_cpu_ht2_t *ht = (orc_cpu_ht2_t *)malloc(sizeof(*ht));
  ht->o2n = (orc_bucket2_t *)malloc(sizeof(orc_bucket2_t) * (num_node + 1));
  ht->n2o = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_node + 1));
  ht->num_node = num_node;
  ht->num_items = 0;
  ht->version = 0;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < num_node; ++i) { /* InitTable, :193-203 */
    ht->o2n[i].key = ORC_EMPTY_KEY;
    ht->o2n[i].local = ORC_EMPTY_KEY;
    ht->o2n[i].index = ORC_EMPTY_KEY;
    ht->o2n[i].version = ORC_EMPTY_KEY;
  }
  return ht;
}

void orc_cpu_ht2_destroy(orc_cpu_ht2_t *ht) {
  if (!ht) return;
  free(ht->o2n);
  free(ht->n2o);
  free(ht);
}

void orc_cpu_ht2_reset(orc_cpu_ht2_t *ht, int threads) { /* :183-191 */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < ht->num_items; ++i) ht->o2n[ht->n2o[i]].key = ORC_EMPTY_KEY;
  ht->num_items = 0;
  ht->version = 0;
}

size_t orc_cpu_ht2_populate(orc_cpu_ht2_t *ht, const orc_id_t *input, size_t num_input, int threads) { /* :53-107 */
  orc_bucket2_t *o2n = ht->o2n;
  const orc_id_t version = ht->version;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < num_input; ++i) {
    const orc_id_t id = input[i];
    const orc_id_t key = __sync_val_compare_and_swap(&o2n[id].key, ORC_EMPTY_KEY, id);
    if (key == ORC_EMPTY_KEY) {
      o2n[id].index = (orc_id_t)i;
      o2n[id].version = version;
    }
  }
  size_t *prefix = (size_t *)calloc((size_t)threads + 1, 16 * sizeof(size_t)); /* one cache line per thread */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < num_input; ++i) {
    const orc_bucket2_t *b = &o2n[input[i]];
    if (b->index == (orc_id_t)i && b->version == version) prefix[16 * (size_t)omp_get_thread_num()]++;
  }
  size_t sum = 0;
  for (int t = 0; t <= threads; ++t) {
    const size_t tmp = prefix[16 * (size_t)t];
    prefix[16 * (size_t)t] = sum;
    sum += tmp;
  }
  const size_t start = ht->num_items;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < num_input; ++i) {
    const orc_id_t id = input[i];
    orc_bucket2_t *b = &o2n[id];
    if (b->index == (orc_id_t)i && b->version == version) {
      const size_t new_id = start + prefix[16 * (size_t)omp_get_thread_num()]++;
      b->local = (orc_id_t)new_id;
      ht->n2o[new_id] = id;
    }
  }
  ht->num_items += prefix[16 * (size_t)threads];
  ht->version++;
  free(prefix);
  return ht->num_items;
}

const orc_id_t *orc_cpu_ht2_unique(const orc_cpu_ht2_t *ht) { return ht->n2o; }

void orc_cpu_ht2_map_edges(const orc_cpu_ht2_t *ht, const orc_id_t *src, const orc_id_t *dst, size_t len,
                           orc_id_t *new_src, orc_id_t *new_dst, int threads) { /* :148-160 */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (size_t i = 0; i < len; ++i) {
    new_src[i] = ht->o2n[src[i]].local;
    new_dst[i] = ht->o2n[dst[i]].local;
  }
}

/* ======================================================================= */
/* Multi-layer loop: dist_loops.cc:62-368 (GPU), cpu_loops.cc:55-192 (CPU) */
/* ======================================================================= */
orc_sample_result_t *orc_do_sample(int sample_type, const orc_id_t *indptr,
                                   const orc_id_t *indices, size_t num_node,
                                   const orc_id_t *seeds, size_t num_seeds,
                                   const size_t *fanouts, size_t num_layer,
                                   orc_xorwow_t *states, size_t num_states) {
  return orc_do_sample_ex(sample_type, indptr, indices, num_node, seeds, num_seeds, fanouts, num_layer, states,
                          num_states, NULL);
}

orc_sample_result_t *orc_do_sample_ex(int sample_type, const orc_id_t *indptr,
                                      const orc_id_t *indices, size_t num_node,
                                      const orc_id_t *seeds, size_t num_seeds,
                                      const size_t *fanouts, size_t num_layer,
                                      orc_xorwow_t *states, size_t num_states,
                                      const orc_sample_extra_t *extra) {
  orc_sample_result_t *r = (orc_sample_result_t *)calloc(1, sizeof(*r));
  r->num_layer = num_layer;
  r->num_src = (size_t *)calloc(num_layer, sizeof(size_t));
  r->num_dst = (size_t *)calloc(num_layer, sizeof(size_t));
  r->num_edge = (size_t *)calloc(num_layer, sizeof(size_t));
  r->row = (orc_id_t **)calloc(num_layer, sizeof(orc_id_t *));
  r->col = (orc_id_t **)calloc(num_layer, sizeof(orc_id_t *));
  r->data = (orc_id_t **)calloc(num_layer, sizeof(orc_id_t *));
  size_t cap = orc_predict_num_nodes(num_seeds, fanouts, num_layer);
  orc_hashtable_t *ht = orc_ht_create(num_node, cap);
  orc_ht_fill_with_duplicates(ht, seeds, num_seeds);
  orc_id_t *cur = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_seeds + 1));
  memcpy(cur, seeds, sizeof(orc_id_t) * num_seeds);
  size_t num_cur = num_seeds;
  for (int i = (int)num_layer - 1; i >= 0; --i) {
    const size_t fanout = fanouts[i];
    orc_id_t *out_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_cur * fanout + 1));
    orc_id_t *out_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_cur * fanout + 1));
    orc_id_t *out_data = NULL;
    size_t num_out = 0;
    switch (sample_type) {
      case ORC_WEIGHTED_KHOP:
        orc_sample_weighted_khop(indptr, indices, extra->prob_table, extra->alias_table, cur, num_cur, fanout, states,
                                 num_states, out_src, out_dst, &num_out);
        break;
      case ORC_WEIGHTED_KHOP_PREFIX:
        orc_sample_weighted_khop_prefix(indptr, indices, extra->prob_table, cur, num_cur, fanout, states, num_states,
                                        out_src, out_dst, &num_out);
        break;
      case ORC_WEIGHTED_KHOP_HASH_DEDUP:
        orc_sample_weighted_khop_hash_dedup(indptr, indices, extra->prob_table, extra->alias_table, cur, num_cur,
                                            fanout, states, num_states, out_src, out_dst, &num_out);
        break;
      case ORC_RANDOM_WALK: /* fanout = num_neighbor = K (operation.cc:174) */
        out_data = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_cur * fanout + 1));
        orc_sample_random_walk(indptr, indices, cur, num_cur, extra->walk_length, extra->restart_prob,
                               extra->num_walk, fanout, states, num_states, out_src, out_dst, out_data, &num_out);
        break;
      case ORC_KHOP3:
        orc_sample_khop3(indptr, indices, cur, num_cur, fanout, states, num_states, out_src, out_dst, &num_out);
        break;
      case ORC_KHOP1:
        orc_sample_khop1(indptr, indices, cur, num_cur, fanout, states, num_states, out_src, out_dst, &num_out);
        break;
      case ORC_KHOP2:
        orc_sample_khop2(indptr, (orc_id_t *)indices, cur, num_cur, fanout, states, num_states, out_src, out_dst,
                         &num_out); /* const_cast, cuda_loops.cc:163 */
        break;
      case ORC_KHOP0:
        orc_sample_khop0(indptr, indices, cur, num_cur, fanout, out_src, out_dst, &num_out);
        break;
      case ORC_CPU_KHOP0:
        orc_cpu_sample_khop0(indptr, indices, cur, num_cur, out_src, out_dst, &num_out, fanout, 1);
        break;
      default:
        assert(0);
    }
    size_t num_unique = orc_ht_fill_with_duplicates(ht, out_dst, num_out);
    orc_id_t *new_src = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_out + 1));
    orc_id_t *new_dst = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_out + 1));
    orc_ht_map_edges(ht, out_src, out_dst, num_out, new_src, new_dst);
    r->num_src[i] = num_unique;
    r->num_dst[i] = num_cur;
    r->num_edge[i] = num_out;
    r->col[i] = new_src; /* train_graph->col = new_src (dist_loops.cc:306) */
    r->row[i] = new_dst; /* train_graph->row = new_dst (:310)             */
    r->data[i] = out_data; /* train_graph->data (:314-319)               */
    free(out_src);
    free(out_dst);
    free(cur);
    cur = (orc_id_t *)malloc(sizeof(orc_id_t) * (num_unique + 1));
    memcpy(cur, orc_ht_unique(ht), sizeof(orc_id_t) * num_unique);
    num_cur = num_unique;
  }
  r->input_nodes = cur;
  r->num_input_nodes = num_cur;
  orc_ht_destroy(ht);
  return r;
}

void orc_sample_result_free(orc_sample_result_t *r) {
  if (!r) return;
  for (size_t i = 0; i < r->num_layer; ++i) { free(r->row[i]); free(r->col[i]); free(r->data[i]); }
  free(r->row); free(r->col); free(r->data);
  free(r->num_src); free(r->num_dst); free(r->num_edge);
  free(r->input_nodes);
  free(r);
}

/* ======================================================================= */
/* Feature cache                                                           */
/* ======================================================================= */
/* cuda_cache_manager_host.cc:61-130 (plain) and :133-254 (partition) */
void orc_cache_build(const orc_id_t *rank_nodes, size_t num_nodes,
                     size_t num_cached, int partition_shuffle,
                     orc_id_t *rank_out, orc_id_t *table) {
  memcpy(rank_out, rank_nodes, sizeof(orc_id_t) * num_nodes);
  if (partition_shuffle) {
    /* std::mt19937 eg(real_num_cached_node); std::shuffle(prefix) :169-171 */
    orc_mt19937_t g;
    orc_mt19937_seed(&g, (uint32_t)num_cached);
    orc_mt19937_shuffle_u32(&g, rank_out, num_cached);
  }
  for (size_t i = 0; i < num_nodes; ++i) table[i] = ORC_EMPTY_KEY;
  for (size_t i = 0; i < num_cached; ++i) table[rank_out[i]] = (orc_id_t)i;
}

/* cuda_cache_manager_device.cu:40-169 + host :355-441: stable partition */
void orc_get_miss_cache_index(const orc_id_t *table, const orc_id_t *nodes,
                              size_t num_nodes, orc_id_t *miss_src,
                              orc_id_t *miss_dst, size_t *num_miss,
                              orc_id_t *hit_src, orc_id_t *hit_dst,
                              size_t *num_hit) {
  size_t m = 0, h = 0;
  for (size_t i = 0; i < num_nodes; ++i) {
    orc_id_t slot = table[nodes[i]];
    if (slot == ORC_EMPTY_KEY) {
      miss_dst[m] = (orc_id_t)i;  /* row in the batch output   */
      miss_src[m] = nodes[i];     /* global id (host row)      */
      ++m;
    } else {
      hit_dst[h] = (orc_id_t)i;
      hit_src[h] = slot;          /* cache slot                */
      ++h;
    }
  }
  *num_miss = m;
  *num_hit = h;
}

/* combine_cache_data :254-275, extract_miss_data :233-252 (mask = all ones),
 * combine_miss_data :209-231 (src_index == NULL -> row i) */
void orc_gather_scatter(void *out, const void *src, const orc_id_t *src_index,
                        const orc_id_t *dst_index, size_t n, size_t row_bytes) {
  for (size_t i = 0; i < n; ++i) {
    size_t s = src_index ? src_index[i] : i;
    size_t d = dst_index ? dst_index[i] : i;
    memcpy((char *)out + d * row_bytes, (const char *)src + s * row_bytes, row_bytes);
  }
}

/* combine_cache_data_for_partition :277-299 + DeviceDistFeature::Get
 * (dist_graph.h:191-205) */
void orc_gather_scatter_partition(void *out, const void *const *parts,
                                  size_t num_part, const orc_id_t *src_index,
                                  const orc_id_t *dst_index, size_t n,
                                  size_t row_bytes) {
  for (size_t i = 0; i < n; ++i) {
    size_t slot = src_index[i];
    size_t part = slot % num_part, real = slot / num_part;
    size_t d = dst_index ? dst_index[i] : i;
    memcpy((char *)out + d * row_bytes, (const char *)parts[part] + real * row_bytes, row_bytes);
  }
}

/* ======================================================================= */
/* GGMS sharding                                                           */
/* ======================================================================= */
/* dist_graph.cu:228-272 */
void orc_partition_graph(const orc_id_t *indptr, const orc_id_t *indices,
                         orc_id_t part_id, orc_id_t num_part,
                         orc_id_t num_part_node, orc_id_t *part_indptr,
                         orc_id_t *part_indices, size_t *indptr_size,
                         size_t *indices_size) {
  size_t edge_count = 0;
  for (orc_id_t i = part_id; i < num_part_node; i += num_part)
    edge_count += indptr[i + 1] - indptr[i];
  size_t isz = num_part_node / num_part + (part_id < num_part_node % num_part ? 1 : 0) + 1;
  if (indptr_size) *indptr_size = isz;
  if (indices_size) *indices_size = edge_count;
  if (!part_indptr || !part_indices) return;
  orc_id_t cnt = 0;
  for (orc_id_t i = part_id; i < num_part_node; i += num_part) {
    orc_id_t ne = indptr[i + 1] - indptr[i];
    orc_id_t real_id = i / num_part;
    part_indptr[real_id] = cnt;
    memcpy(&part_indices[cnt], &indices[indptr[i]], ne * sizeof(orc_id_t));
    cnt += ne;
  }
  part_indptr[isz - 1] = cnt;
}

/* dist_graph.cu:493-521 */
size_t orc_partition_feature(const void *feat, size_t row_bytes,
                             const orc_id_t *rank_nodes, orc_id_t num_cache,
                             orc_id_t part_id, orc_id_t num_part, void *out) {
  size_t cnt = 0;
  for (orc_id_t i = part_id; i < num_cache; i += num_part) {
    if (out)
      memcpy((char *)out + cnt * row_bytes, (const char *)feat + (size_t)rank_nodes[i] * row_bytes, row_bytes);
    ++cnt;
  }
  return cnt;
}

/* dist_engine.cc:223-232 */
orc_id_t orc_num_cache_node(const orc_id_t *indptr, orc_id_t num_node,
                            double percentage) {
  orc_id_t num_edge = indptr[num_node];
  orc_id_t num_cache_edge = (orc_id_t)(num_edge * percentage);
  orc_id_t n = 0;
  while (n < num_node && indptr[n] < num_cache_edge) ++n;
  return n;
}
