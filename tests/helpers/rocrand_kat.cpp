// Prints XORWOW known answers from rocRAND's own host engine (third-party dependency the path
// uses for its random stream).  Built by tests/test_rng.py with g++; not part of the product.
#include <cstdio>
#include <cstdlib>
#define __HIP_PLATFORM_AMD__ 1
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>

int main(int argc, char **argv)
{
    for (int i = 1; i + 2 < argc; i += 3) {
        const unsigned long long seed = strtoull(argv[i], nullptr, 10);
        const unsigned long long sub = strtoull(argv[i + 1], nullptr, 10);
        const unsigned long long off = strtoull(argv[i + 2], nullptr, 10);
        rocrand_state_xorwow st;
        rocrand_init(seed, sub, off, &st);
        for (int k = 0; k < 8; ++k) {
            printf("%08x ", rocrand(&st));
        }
        printf("%.9g\n", rocrand_uniform(&st));
    }
    return 0;
}
