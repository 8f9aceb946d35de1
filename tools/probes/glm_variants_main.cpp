#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define DECL(n) extern "C" float run_##n(const float*, const float*, const float*, const float*, int64_t, int, int, float*, int, int);
DECL(base) DECL(noelem) DECL(nosplitg) DECL(nogemm2) DECL(noxb) DECL(nogemm1) DECL(nosplitx) DECL(nosplit) DECL(sched6) DECL(sched8) DECL(sched10) DECL(sched11) DECL(sched13) DECL(minimal) DECL(nomfma) DECL(novalu)
namespace pa { int cu_count() { return 256; } }
int main(int argc, char** argv) {
  const int64_t N = 1000000; const int D = 32, P = 64;
  std::vector<float> hX(N * D), hy(N), hw(P * D), hb(P);
  srand(1);
  for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 3.4f;
  for (auto& v : hy) v = rand() & 1;
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
  for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
  float *X, *y, *w, *b, *part;
  (void)hipMalloc(&X, hX.size() * 4); (void)hipMalloc(&y, hy.size() * 4); (void)hipMalloc(&w, hw.size() * 4); (void)hipMalloc(&b, hb.size() * 4);
  (void)hipMalloc(&part, (size_t)1024 * 2176 * 4 + 16000000);
  (void)hipMemcpy(X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  for (int nb : {512}) {
    printf("nblocks=%d\n", nb);
#define RUN(n) printf("  %-10s %8.1f us\n", #n, run_##n(X, y, w, b, N, D, P, part, nb, 20));
    RUN(base) RUN(sched6) RUN(sched8) RUN(sched10) RUN(sched11) RUN(sched13) RUN(base) RUN(noelem) RUN(nosplitg) RUN(nosplitx) RUN(nosplit) RUN(nogemm2) RUN(noxb) RUN(nogemm1) RUN(nomfma) RUN(novalu) RUN(minimal)
  }
  return 0;
}
