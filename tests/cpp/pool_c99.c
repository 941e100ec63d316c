/* include/cilqr.h is a C header: this file is compiled as C99 (gcc -std=c99 -pedantic -Wall -Werror) and linked against the
 * C-ABI library alone.  It walks the cilqr_pool_* calls the way INTEGRATION.md section 4 shows them; without a GPU only the
 * argument checks run (exit code 0 = every call answered as the header says). */
#include <stdio.h>

#include "cilqr.h"

int main(void) {
  cilqr_config cfg;
  cilqr_pool_handle pool = 0;
  if (cilqr_default_config(&cfg, 50) != CILQR_OK) return 1;
  if (cilqr_abi_version() != CILQR_ABI_VERSION) return 2;
  if (cilqr_pool_create(&cfg, 0, 0, 8, 16, 64, &pool) != CILQR_ERR_ARG || pool != 0) return 3;      /* no handles */
  if (cilqr_pool_create(0, 0, 3, 8, 16, 64, &pool) != CILQR_ERR_NULL) return 4;
  if (cilqr_pool_submit(0, 0, 0) != CILQR_ERR_NULL || cilqr_pool_wait(0) != CILQR_ERR_NULL) return 5;
  if (cilqr_pool_depth(0) != 0 || cilqr_pool_device_bytes(0) != 0 || cilqr_pool_handle_at(0, 0) != 0) return 6;
  if (cilqr_pool_set_option(0, CILQR_OPT_FINISH_THRESHOLD, 0) != CILQR_ERR_NULL) return 7;
  if (cilqr_pool_destroy(0) != CILQR_ERR_NULL) return 8;
  printf("pool_c99 ok (ABI %d)\n", cilqr_abi_version());
  return 0;
}
