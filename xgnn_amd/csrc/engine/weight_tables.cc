// weight_tables.cc -- per-edge tables of the weighted samplers, built on the host from per-edge weights.
//
// Reference: utility/data-process/toolkit/weight/create_alias_table.cc:105-170 (prob_table.bin / alias_table.bin,
// Vose's alias method per neighbour list, float arithmetic, FIFO work lists, the alias slot holds the GLOBAL node id
// of the donor neighbour) and create_prob_prefix_table.cc:94-123 (prob_prefix_table.bin, running float sum per list).
// The reference draws the weights inside the tool (unseeded); here they are an input, so datasets are reproducible.
#include <algorithm>
#include <cstdint>
#include <deque>
#include <thread>
#include <vector>

#include "../../../include/ggms.h"

namespace {

template <typename F>
void parallel_nodes(size_t num_node, int num_threads, F body) {
  const size_t nt = (size_t)std::max(1, num_threads);
  if (nt == 1 || num_node < 4096) {
    body(0, num_node);
    return;
  }
  std::vector<std::thread> pool;
  const size_t chunk = (num_node + nt - 1) / nt;
  for (size_t t = 0; t < nt; ++t) {
    const size_t lo = t * chunk, hi = std::min(num_node, lo + chunk);
    if (lo < hi) pool.emplace_back([=] { body(lo, hi); });
  }
  for (auto &th : pool) th.join();
}

} // namespace

extern "C" {

int ggms_build_alias_table_host(const ggms_id_t *indptr, const ggms_id_t *indices, size_t num_node,
                                const float *weights, float *prob_table, ggms_id_t *alias_table, int num_threads) {
  if (!indptr || (num_node && (!indices || !weights || !prob_table || !alias_table))) return GGMS_ERR_INVALID;
  parallel_nodes(num_node, num_threads, [=](size_t lo, size_t hi) {
    std::vector<float> w;
    std::deque<uint32_t> smalls, larges;
    for (size_t v = lo; v < hi; ++v) {
      const uint32_t off = indptr[v], len = indptr[v + 1] - off;
      w.assign(weights + off, weights + off + len);
      float sum = 0.0f;
      for (uint32_t i = 0; i < len; ++i) sum += w[i]; // :128 (sequential float sum)
      for (uint32_t i = 0; i < len; ++i) {            // :131-135
        w[i] /= sum;
        w[i] *= (float)len;
      }
      smalls.clear();
      larges.clear();
      for (uint32_t i = 0; i < len; ++i) (w[i] < 1.0f ? smalls : larges).push_back(i); // :141-147
      while (!smalls.empty() && !larges.empty()) { // :149-166
        const uint32_t s = smalls.front(), l = larges.front();
        smalls.pop_front();
        larges.pop_front();
        prob_table[off + s] = w[s];
        alias_table[off + s] = indices[off + l];
        w[l] -= (1 - w[s]);
        (w[l] < 1.0f ? smalls : larges).push_back(l);
      }
      // what is left takes its own neighbour with probability 1; its alias slot is never read (the tool leaves the
      // vector's zero there, :212)
      for (uint32_t i : larges) { prob_table[off + i] = 1.0f; alias_table[off + i] = 0; }
      for (uint32_t i : smalls) { prob_table[off + i] = 1.0f; alias_table[off + i] = 0; }
    }
  });
  return GGMS_OK;
}

int ggms_build_prob_prefix_table_host(const ggms_id_t *indptr, size_t num_node, const float *weights,
                                      float *prob_prefix_table, int num_threads) {
  if (!indptr || (num_node && (!weights || !prob_prefix_table))) return GGMS_ERR_INVALID;
  parallel_nodes(num_node, num_threads, [=](size_t lo, size_t hi) {
    for (size_t v = lo; v < hi; ++v) {
      const uint32_t off = indptr[v], len = indptr[v + 1] - off;
      float sum = 0.0f;
      for (uint32_t i = 0; i < len; ++i) { // create_prob_prefix_table.cc:104-122
        sum += weights[off + i];
        prob_prefix_table[off + i] = sum;
      }
    }
  });
  return GGMS_OK;
}

} // extern "C"
