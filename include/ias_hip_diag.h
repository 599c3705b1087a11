/* Entry points that exist ONLY in the diagnostic library (csrc/libias_hip_diag.so: the product sources compiled with
 * -DIAS_DIAG).  The product library (csrc/libias_hip.so, include/ias_hip.h) reads nothing from the environment, keeps no
 * mutable global state and ships one kernel per operation and shape; the diagnostic library additionally
 *   - honours the IAS_* environment switches of scripts/diag (superseded kernels, grid sizes, ...),
 *   - contains the kernels only such a switch reaches (the matrix-core STFT of csrc/stft_mfma_kernels.hip, the LDS form of
 *     the stem weight gradient),
 *   - exports the process-wide switch below.
 * It is loaded by scripts/diag, by bench.py for its `roofline.dxd` comparison figure, and by the tests that compare a
 * superseded kernel with its replacement (inverse-audio-synthesis_amd/_lib.py: load_diag, use_library) -- never by the
 * package itself. */
#ifndef IAS_HIP_DIAG_H
#define IAS_HIP_DIAG_H
#include "ias_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* form 1: the covariance term of ias_vicreg_loss / _backward from the batch side wherever the shape allows (the product
 * library's rule); 0: always the D x D kernels of rounds 1-3 (the reference's literal order, vicreg.py:47-51); -1: default
 * (1 unless the environment has IAS_VICREG_DXD=1).  Process-wide in THIS library; forward and backward of one loss must
 * run under the same setting. */
int ias_vicreg_set_form(int form);

#ifdef __cplusplus
}
#endif
#endif
