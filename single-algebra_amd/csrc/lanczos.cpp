#include "lanczos.h"

namespace sapca {

template <typename T>
void lanczos_fit(sapca_handle_s& h) {
  (void)h;
  throw Error(SAPCA_ERR_SVD, "SVD computation failed: Lanczos path not built yet");
}

template void lanczos_fit<float>(sapca_handle_s&);
template void lanczos_fit<double>(sapca_handle_s&);

}  // namespace sapca
