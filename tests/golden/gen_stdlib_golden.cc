// Generates golden vectors for the libstdc++ constructs the reference's
// shufflers / cache-rank shuffle rely on (third-party dependency: GCC's
// libstdc++, whose uniform_int_distribution / std::shuffle algorithms are
// implementation-defined).  Our own code; the expressions mirror the call
// sites dist/dist_shuffler_aligned.cc:94-99 and
// cuda/cuda_cache_manager_host.cc:169-171.
//   g++ -O2 -o gen gen_stdlib_golden.cc && ./gen > stdlib_golden.txt
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

static void fisher_yates(std::vector<uint32_t> &d, uint64_t seed) {
  auto g = std::default_random_engine(seed);
  size_t n = d.size();
  for (size_t i = 0; n && i < n - 1; i++) {
    std::uniform_int_distribution<size_t> dist(i, n - 1);
    size_t c = dist(g);
    std::swap(d[i], d[c]);
  }
}

int main() {
  // minstd Fisher-Yates, sizes x seeds
  for (size_t n : {1, 2, 7, 100, 1000}) {
    for (uint64_t seed : {0ull, 1ull, 5ull, 2147483647ull, 123456789012ull}) {
      std::vector<uint32_t> d(n);
      for (size_t i = 0; i < n; i++) d[i] = (uint32_t)(i * 3 + 1);
      fisher_yates(d, seed);
      printf("minstd %zu %llu", n, (unsigned long long)seed);
      for (auto v : d) printf(" %u", v);
      printf("\n");
    }
  }
  // std::shuffle with mt19937(seed = n) over a prefix, both code paths
  // (pairwise draws when n*n <= 2^32-1, one draw per element otherwise)
  for (size_t n : {0, 1, 2, 3, 10, 11, 1000, 65535, 65536, 70001}) {
    std::vector<uint32_t> d(n + 5);
    for (size_t i = 0; i < d.size(); i++) d[i] = (uint32_t)(i ^ 0x5a5a);
    std::mt19937 eg(n);
    std::shuffle(d.begin(), d.begin() + n, eg);
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the words
    for (auto v : d) { h ^= v; h *= 1099511628211ull; }
    printf("mtshuffle %zu %llu", n, (unsigned long long)h);
    for (size_t i = 0; i < d.size() && i < 16; i++) printf(" %u", d[i]);
    printf("\n");
  }
  return 0;
}
