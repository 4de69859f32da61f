// rocrand_kat.cpp -- TEST INFRASTRUCTURE (oracle/): prints the first 16 outputs of rocRAND's own
// Philox engine, rocrand_device::philox4x32_10_engine(seed, subsequence, offset), compiled for the
// HOST from /opt/rocm/include/rocrand/rocrand_philox4x32_10.h (the engine is __host__ __device__).
// tests/test_oracle_golden.py compares them with tests/golden/philox_kat.json "streams", which is
// what pins the RNG contract of include/msnake.h (draw i of env g = engine(seed, g, i)) to rocRAND
// itself and not only to this repo's restatement.
//   usage: rocrand_kat seed subsequence offset [seed subsequence offset ...]
#include <rocrand/rocrand_philox4x32_10.h>

#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    for (int i = 1; i + 2 < argc; i += 3) {
        const unsigned long long seed = strtoull(argv[i], nullptr, 0), sub = strtoull(argv[i + 1], nullptr, 0),
                                 off = strtoull(argv[i + 2], nullptr, 0);
        rocrand_device::philox4x32_10_engine eng(seed, sub, off);
        for (int k = 0; k < 16; ++k) printf("%u%c", eng(), k == 15 ? '\n' : ' ');
    }
    return 0;
}
