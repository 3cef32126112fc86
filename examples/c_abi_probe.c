/* Plain-C consumer of libuvtg.so (no Python, no torch, no HIP headers): includes include/uvtg.h, links the library and walks the
 * size / parameter-table queries a host program needs before it allocates device memory.  Proves that the boundary is a C ABI a
 * maintainer can bind from any language; compiled and run by tests/test_capi.py (no GPU needed: the
 * queries are host arithmetic).
 *
 *     gcc -std=c99 -Iinclude examples/c_abi_probe.c -o /tmp/c_abi_probe -Lunivtg_amd -luvtg -Wl,-rpath,$PWD/univtg_amd
 */
#include <stdio.h>
#include <string.h>
#include "uvtg.h"

int main(void) {
  uvtg_dims d;
  memset(&d, 0, sizeof d);
  d.struct_size = (int)sizeof d;
  d.B = 256; d.Lv = 75; d.Lt = 32; d.d = 1024; d.H = 8; d.F = 1024; d.E = 4; d.Dv = 2818; d.Dt = 512; d.n_proj = 2;
  d.training = 1; d.proj_precise = 1; d.p_in = 0.5f; d.p_path = 0.1f; d.seed = 1; d.max_q_l = 75;
  const int n = uvtg_param_count(&d);
  long long off[256];
  if (n <= 0 || n >= 255 || uvtg_param_offsets(&d, off) != 0) { printf("param table: error\n"); return 1; }
  printf("version %d params %d elements %lld workspace_bytes %zu wcache_bytes %zu\n", uvtg_version(), n, off[n],
         uvtg_workspace_bytes(&d), uvtg_wcache_bytes(&d));
  d.n_proj = 3; d.use_txt_pos = 1;
  printf("n_proj=3 use_txt_pos=1 params %d\n", uvtg_param_count(&d));
  d.struct_size = 4;                                  /* a caller built against another header: refused, not misread */
  printf("bad struct_size -> %d (%s)\n", uvtg_param_offsets(&d, off), uvtg_strerror(uvtg_param_offsets(&d, off)));
  return 0;
}
