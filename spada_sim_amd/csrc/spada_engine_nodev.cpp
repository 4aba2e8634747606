// CPU SANITIZER BUILD ONLY (make -C spada_sim_amd/csrc asan): the device half of libspada_spgemm.so replaced by entry points that
// refuse -- spada_create reports SPADA_ERR_NO_DEVICE, everything that needs a context SPADA_ERR_STATE -- so that the host half
// (spada_host.cpp, spada_cycle.cpp) can be loaded and exercised under -fsanitize=address,undefined by the non-GPU test set.
// This file is never part of the product library and computes nothing: there is no CPU SpGEMM path in this repository.
#include "spada_internal.hpp"

using namespace spada;

#define NODEV(name) return fail(SPADA_ERR_STATE, name ": no engine context (sanitizer build of the host half: no device code)")

extern "C" {

int spada_device_count(void) { return 0; }
int spada_create(const spada_options *, spada_ctx **out)
{
    if (out) *out = nullptr;
    return fail(SPADA_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU path (sanitizer build of the host half)");
}
void spada_destroy(spada_ctx *) {}
int spada_spgemm_symbolic(spada_ctx *, const spada_csr_view *, const spada_csr_view *, uint64_t *) { NODEV("spada_spgemm_symbolic"); }
int spada_spgemm_numeric(spada_ctx *, uint64_t *, uint64_t *, double *) { NODEV("spada_spgemm_numeric"); }
int spada_spgemm_symbolic_reordered(spada_ctx *, const spada_csr_view *, const spada_csr_view *, int, uint64_t *, uint64_t *)
{
    NODEV("spada_spgemm_symbolic_reordered");
}
int spada_spgemm_fused(spada_ctx *, const spada_csr_view *, const spada_csr_view *, uint64_t, uint64_t *, uint64_t *, double *, uint64_t *)
{
    NODEV("spada_spgemm_fused");
}
int spada_dev_csr_upload(spada_ctx *, const spada_csr_view *, spada_dev_csr **) { NODEV("spada_dev_csr_upload"); }
void spada_dev_csr_free(spada_ctx *, spada_dev_csr *) {}
int spada_dev_csr_reorder(spada_ctx *, const spada_dev_csr *, const spada_dev_csr *, int, spada_dev_csr **) { NODEV("spada_dev_csr_reorder"); }
int spada_dev_csr_rowmap(spada_ctx *, const spada_dev_csr *, uint64_t *) { NODEV("spada_dev_csr_rowmap"); }
int spada_dev_unpermute_c(spada_ctx *, const spada_dev_csr *, const void *, const void *, const void *, void *, void *, void *)
{
    NODEV("spada_dev_unpermute_c");
}
int spada_dev_spgemm_symbolic(spada_ctx *, const spada_dev_csr *, const spada_dev_csr *, uint64_t, uint64_t, uint64_t *)
{
    NODEV("spada_dev_spgemm_symbolic");
}
int spada_dev_spgemm_numeric(spada_ctx *, void *, void *, void *) { NODEV("spada_dev_spgemm_numeric"); }
int spada_dev_spgemm_numeric_plan(spada_ctx *, uint32_t, uint64_t *) { NODEV("spada_dev_spgemm_numeric_plan"); }
int spada_dev_spgemm_numeric_chunk(spada_ctx *, uint32_t, void *, void *, void **) { NODEV("spada_dev_spgemm_numeric_chunk"); }
int spada_dev_spgemm_indptr(spada_ctx *, void *) { NODEV("spada_dev_spgemm_indptr"); }
int spada_dev_synchronize(spada_ctx *) { NODEV("spada_dev_synchronize"); }
int spada_dev_spgemm_fused(spada_ctx *, const spada_dev_csr *, const spada_dev_csr *, uint64_t, uint64_t, void *, void *, void *, uint64_t,
                           uint64_t *)
{
    NODEV("spada_dev_spgemm_fused");
}
int spada_dev_spgemm_fused_owned(spada_ctx *, const spada_dev_csr *, const spada_dev_csr *, uint64_t, uint64_t, uint64_t, void **, void **,
                                 void **, uint64_t *)
{
    NODEV("spada_dev_spgemm_fused_owned");
}
int spada_dev_spgemm_numeric_owned(spada_ctx *, void **, void **, void **) { NODEV("spada_dev_spgemm_numeric_owned"); }
int spada_dev_download_c(spada_ctx *, const void *, const void *, const void *, uint64_t, uint64_t, uint64_t *, uint64_t *, double *)
{
    NODEV("spada_dev_download_c");
}
int spada_get_stats(const spada_ctx *, spada_stats *) { return fail(SPADA_ERR_INVALID, "spada_get_stats: null argument"); }
int spada_set_phase_timing(spada_ctx *, int) { return fail(SPADA_ERR_INVALID, "spada_set_phase_timing: null context"); }

}  // extern "C"
