/* A plain-C client of the C ABI: proves include/jacobiforcing.h is valid C (no C++/torch types in the signatures) and that the
 * shared library links and answers without a GPU.  Built and run by tests/test_kernels.py::test_plain_c_client. */
#include <stdio.h>
#include <string.h>

#include "jacobiforcing.h"

int main(void) {
    jf_mb_params p;
    memset(&p, 0, sizeof p);
    p.n = 32; p.K = 2; p.spawn_threshold = 28; p.pool_size = 4; p.eos_id = -1; p.pad_id = 0; p.max_iter = 128; p.max_blocks = 3;
    p.lookahead_start_ratio = 0.0;
    printf("version=%d state_ints=%lld max_rows=%d max_tokens=%d desc=%zu params=%zu\n", jf_version(),
           (long long)jf_mb_state_ints(&p), (int)jf_mb_max_rows(&p), (int)jf_mb_max_tokens(&p), sizeof(jf_mb_desc), sizeof(jf_mb_params));
    /* argument validation happens before any HIP call: a NULL buffer is JF_E_INVALID with a message */
    int rc = jf_argmax_partial(NULL, JF_BF16, 4, 152064, 152064, NULL, NULL);
    printf("rc=%d err=%s\n", rc, jf_last_error());
    return (rc == JF_E_INVALID && jf_version() == JF_VERSION) ? 0 : 1;
}
