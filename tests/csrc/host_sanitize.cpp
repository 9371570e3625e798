// CPU sanitizer harness for the host-only pieces of the C ABI (remixt_amd/csrc/rmx_host.h).  Built by
// tests/test_sanitizers_cpu.py with g++ -fsanitize=address,undefined -fno-sanitize-recover=all and run as a child
// process; it prints one line per case that the Python side compares with numpy / scipy / remixt_amd.lockstep.fmin_1d.
//   host_sanitize nm <case> <x0>         Nelder-Mead on objective <case>: every requested point, then "xopt <x> fcalls <n>"
//   host_sanitize ws <seed> <n> <k>      weighted_search on a seeded weight vector (zeros included): the k indices
//   host_sanitize wr <seed> <n> <size> <k>  weighted_sample_round on a strided weight column until <size> distinct indices are found
//   host_sanitize cc <seed> <N> <S> <M>  compress_cn_states on a seeded table sequence: classes, then the class ids
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../remixt_amd/csrc/rmx_host.h"

// objectives with exactly reproducible arithmetic (+, -, *, fabs only; the Python twin repeats them operation by operation)
static double objective(int c, double x) {
    switch (c) {
    case 0: return (x - 1.3) * (x - 1.3) + 0.1 * std::fabs(x);
    case 1: return std::fabs(x - 250.) * 0.01 + 3.;
    case 2: { const double t = x * 0.001 - 0.7; return t * t * t * t - 0.3 * t * t + 0.05 * t; }
    case 3: return (x < 10. || x > 3000.) ? INFINITY : (x - 1999.5) * (x - 1999.5) * 1e-6;     // +inf outside the bounds, cn_model.py:542-543
    default: return 0. * x + 1.;                                                                  // flat: ends on the tolerances at once
    }
}

// splitmix64: the Python side regenerates the same stream
static uint64_t sm_state;
static uint64_t sm_next() { uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double sm_unit() { return (double)(sm_next() >> 11) / 9007199254740992.; }

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "nm") && argc == 4) {
        const int c = atoi(argv[2]);
        const double x0 = atof(argv[3]);
        rmxh::Nm1 nm;
        double f = 0.;
        int guard = 0;
        while (nm.advance(x0, f) && guard++ < 1000) {
            f = objective(c, nm.req);
            printf("req %.17g f %.17g\n", nm.req, f);
        }
        printf("xopt %.17g fcalls %d last %.17g\n", nm.xopt(), nm.fcalls, nm.last);
        return 0;
    }
    if (!strcmp(argv[1], "ws") && argc == 5) {
        sm_state = strtoull(argv[2], 0, 10);
        const int64_t n = atoll(argv[3]);
        const int k = atoi(argv[4]);
        std::vector<double> p((size_t)n), u((size_t)k);
        for (auto &v : p) { const double t = sm_unit(); v = t < 0.3 ? 0. : t; }      // zero weights: masked segments
        p[(size_t)n - 1] = 0.;                                                       // a trailing zero: the clamp to n - 1 must not be needed for u < 1
        for (auto &v : u) v = sm_unit();
        if (k > 0) u[0] = 0.;
        std::vector<int64_t> out((size_t)k);
        int64_t pos = -1;
        const int rc = rmxh::weighted_search(p.data(), n, u.data(), k, out.data(), &pos);
        printf("rc %d positive %lld idx", rc, (long long)pos);
        for (auto v : out) printf(" %lld", (long long)v);
        printf("\n");
        // argument errors must be refused, not dereferenced
        printf("null %d empty %d\n", rmxh::weighted_search(nullptr, n, u.data(), k, out.data(), &pos), rmxh::weighted_search(p.data(), 0, u.data(), k, out.data(), &pos));
        return 0;
    }
    if (!strcmp(argv[1], "wr") && argc == 6) {
        // weighted_sample_round on a strided column with zeros: rounds until `size` distinct indices are found; prints them
        sm_state = strtoull(argv[2], 0, 10);
        const int64_t n = atoll(argv[3]);
        const int size = atoi(argv[4]), k = atoi(argv[5]);
        std::vector<double> q((size_t)n * 2);
        double norm = 0.;
        for (int64_t i = 0; i < n; i++) { const double t = sm_unit(); q[(size_t)i * 2] = -1.; q[(size_t)i * 2 + 1] = t < 0.5 ? 0. : t; }
        for (int64_t i = 0; i < n; i++) norm += q[(size_t)i * 2 + 1];
        std::vector<int64_t> found((size_t)size);      // cap == size: the round must stop writing at the capacity
        int32_t nf = 0, rounds = 0;
        int64_t pos = -1;
        while (nf < size && rounds < 50) {
            std::vector<double> u((size_t)k);
            for (auto &v : u) v = sm_unit();
            const int rc = rmxh::weighted_sample_round(q.data() + 1, n, 2, norm, u.data(), k, found.data(), &nf, size, &pos);
            if (rc) { printf("rc %d\n", rc); return 0; }
            if (pos < size) break;
            rounds++;
        }
        printf("rounds %d positive %lld idx", rounds, (long long)pos);
        for (int i = 0; i < nf; i++) printf(" %lld", (long long)found[(size_t)i]);
        printf("\n");
        int32_t bad = 5;
        printf("args %d %d %d\n", rmxh::weighted_sample_round(nullptr, n, 2, norm, nullptr, 0, found.data(), &nf, size, &pos),
               rmxh::weighted_sample_round(q.data() + 1, n, 2, 0., nullptr, 0, found.data(), &nf, size, &pos),
               rmxh::weighted_sample_round(q.data() + 1, n, 2, norm, nullptr, 0, found.data(), &bad, 4, &pos));
        return 0;
    }
    if (!strcmp(argv[1], "cc") && argc == 6) {
        sm_state = strtoull(argv[2], 0, 10);
        const int N = atoi(argv[3]), S = atoi(argv[4]), M = atoi(argv[5]);
        const size_t tsz = (size_t)S * M * 2;
        const int NT = 3;
        std::vector<int64_t> tables(NT * tsz), dense((size_t)N * tsz);
        for (auto &v : tables) v = (int64_t)(sm_next() % 5);
        std::vector<int> want((size_t)N);
        for (int n = 0; n < N; n++) { want[(size_t)n] = (int)(sm_next() % NT); memcpy(&dense[(size_t)n * tsz], &tables[(size_t)want[(size_t)n] * tsz], tsz * 8); }
        std::vector<int32_t> cls((size_t)N);
        std::vector<int64_t> out(NT * tsz);
        int32_t C = -1;
        int rc = rmxh::compress_cn_states(dense.data(), N, S, M, NT, cls.data(), out.data(), &C);
        int bad = 0;
        for (int n = 0; n < N && !rc; n++) bad += memcmp(&out[(size_t)cls[(size_t)n] * tsz], &dense[(size_t)n * tsz], tsz * 8) != 0;
        printf("rc %d classes %d mismatches %d\n", rc, C, bad);
        // one class too few: refused with RMX_EUNSUPPORTED before anything is written past classes_out
        std::vector<int64_t> small((size_t)(C > 1 ? C - 1 : 1) * tsz);
        rc = rmxh::compress_cn_states(dense.data(), N, S, M, C > 1 ? C - 1 : 1, cls.data(), small.data(), &C);
        printf("too_few rc %d\n", rc);
        return 0;
    }
    return 2;
}
