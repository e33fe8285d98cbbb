/* Plain C against the C ABI (no Python, no torch, no GPU): the header parses as C, the library links, and the host-side entry
 * points behave.  Built and run by tests/test_sampler.py::test_c_program_links_and_runs_against_the_abi. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "drx.h"

int main(void) {
  if (drx_version() != DRX_VERSION) { printf("version mismatch\n"); return 1; }
  if (strlen(drx_strerror(DRX_EINVAL)) == 0 || strlen(drx_strerror(DRX_ERETRY)) == 0) { printf("strerror\n"); return 2; }
  /* random.Random(10).random() twice and randint(0, 9) — values of CPython 3 */
  DrxRng *r = drx_rng_create(10);
  double a = drx_rng_random(r), b = drx_rng_random(r);
  int64_t c = drx_rng_randint(r, 0, 9);
  drx_rng_destroy(r);
  printf("%.17g %.17g %lld\n", a, b, (long long)c);
  /* a 3-row dataset, PointSampler(neg_ratio 1) */
  int32_t uid[3] = {0, 0, 1}, iid[3] = {0, 1, 0};
  double val[3] = {1.0, 1.0, 1.0};
  DrxSampler *s = drx_sampler_create(uid, iid, val, 3, 1, 1, 0.001, 7);
  if (!s) { printf("sampler\n"); return 3; }
  int32_t ou[8], oi[8];
  double ov[8];
  int rc = drx_sampler_sample(s, 8, ou, oi, ov);
  drx_sampler_destroy(s);
  if (rc != DRX_OK) { printf("sample rc %d\n", rc); return 4; }
  for (int i = 0; i < 8; ++i)
    if (ou[i] < 0 || ou[i] > 1 || oi[i] < 0 || oi[i] > 1) { printf("range\n"); return 5; }
  /* a null parameter block is refused, not dereferenced */
  if (drx_cdae_scratch_bytes(NULL, 1, 0, 0) != 0) { printf("scratch\n"); return 6; }
  printf("ok\n");
  return 0;
}
