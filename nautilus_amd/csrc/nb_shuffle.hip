// Host side of the emulator training: the per-epoch minibatch orders.
//
// scikit-learn's MLPRegressor reshuffles its sample index before every epoch
// with sklearn.utils.shuffle(sample_idx, random_state=self._random_state)
// (_multilayer_perceptron.py:700-704), i.e. numpy's legacy
// RandomState.shuffle -- Fisher-Yates from the back, every index drawn by
// masked rejection from 32-bit MT19937 outputs (numpy/random/mtrand.pyx
// `_shuffle_raw`, distributions.c `random_interval`) -- and the new order is
// the old one gathered through that permutation.  The reference reaches it
// through neural.py:79-98.  At config-5 sizes (8 networks x 2 x 10^5 rows) the
// numpy calls cost ~9 ms per network and epoch on one Python thread, more than
// the GPU needs for the epoch itself; here the same streams are advanced by
// one native thread per network, bit for bit (tests/test_host_logic.py
// compares with numpy).  No device code in this file.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "nb_common.h"

namespace {

struct Mt19937 {
  uint32_t* key;   // 624 words, numpy's get_state()[1]
  int pos;         // get_state()[2]

  void refill() {
    constexpr int N = 624, M = 397;
    constexpr uint32_t MATRIX_A = 0x9908b0dfu, UPPER = 0x80000000u,
                       LOWER = 0x7fffffffu;
    int kk = 0;
    uint32_t y;
    for (; kk < N - M; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; kk < N - 1; ++kk) {
      y = (key[kk] & UPPER) | (key[kk + 1] & LOWER);
      key[kk] = key[kk + (M - N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (key[N - 1] & UPPER) | (key[0] & LOWER);
    key[N - 1] = key[M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    pos = 0;
  }

  inline uint32_t next() {
    if (pos >= 624) refill();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
};

// one stream: n_epochs shuffles composed onto `order`, every epoch's order
// written to out[ep * n ...]
void shuffle_stream(uint32_t* key, int32_t* pos, int64_t n, int32_t n_epochs,
                    int32_t* order, int32_t* out) {
  Mt19937 g{key, *pos};
  std::vector<int32_t> perm((size_t)n);
  for (int ep = 0; ep < n_epochs; ++ep) {
    for (int64_t i = 0; i < n; ++i) perm[(size_t)i] = (int32_t)i;
    for (int64_t i = n - 1; i >= 1; --i) {
      // random_interval(bitgen, i): smallest all-ones mask >= i, rejection
      uint32_t mask = (uint32_t)i;
      mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
      mask |= mask >> 8; mask |= mask >> 16;
      uint32_t j;
      while ((j = g.next() & mask) > (uint32_t)i) {}
      const int32_t tmp = perm[j];
      perm[j] = perm[(size_t)i];
      perm[(size_t)i] = tmp;
    }
    int32_t* dst = out + (size_t)ep * (size_t)n;
    for (int64_t i = 0; i < n; ++i) dst[i] = order[perm[(size_t)i]];
    memcpy(order, dst, (size_t)n * sizeof(int32_t));
  }
  *pos = g.pos;
}

}  // namespace

extern "C" {

// Advance n_streams independent MT19937 states (key_of[s]: 624 words, pos[s])
// through n_epochs legacy-numpy shuffles of n_of[s] elements each, composing
// them onto order_of[s] (in/out) and writing the order of every epoch to
// out_of[s] (n_epochs x n_of[s], row-major).  Streams with out_of[s] == NULL
// are skipped.  One host thread per stream.
int nb_host_shuffle_epochs(int32_t n_streams, uint32_t* const* key_of,
                           int32_t* pos, const int64_t* n_of, int32_t n_epochs,
                           int32_t* const* order_of, int32_t* const* out_of) {
  if (n_streams < 0 || n_epochs < 0) {
    nb_set_error("nb_host_shuffle_epochs: bad shape");
    return NB_ERR_ARG;
  }
  for (int s = 0; s < n_streams; ++s)
    if (out_of[s] != nullptr && (n_of[s] < 1 || n_of[s] > 0x7fffffffll ||
                                 pos[s] < 0 || pos[s] > 624)) {
      nb_set_error("nb_host_shuffle_epochs: stream %d has a bad state", s);
      return NB_ERR_ARG;
    }
  std::vector<std::thread> workers;
  int first = -1;
  for (int s = 0; s < n_streams; ++s) {
    if (out_of[s] == nullptr) continue;
    if (first < 0) { first = s; continue; }      // runs on the calling thread
    workers.emplace_back(shuffle_stream, key_of[s], pos + s, n_of[s], n_epochs,
                         order_of[s], out_of[s]);
  }
  if (first >= 0)
    shuffle_stream(key_of[first], pos + first, n_of[first], n_epochs,
                   order_of[first], out_of[first]);
  for (auto& w : workers) w.join();
  return NB_OK;
}

}  // extern "C"
