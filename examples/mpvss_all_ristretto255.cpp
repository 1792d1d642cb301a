// Counterpart of the reference's examples/mpvss_all_ristretto255.rs over the C++ mirror (see mpvss_all_ec.hpp).
//   build: make -C mpvss_rs_amd/csrc examples      run: ./examples/mpvss_all_ristretto255 [seed]
#include "mpvss_all_ec.hpp"

int main(int argc, char** argv) {
  return run_mpvss_all<mpvss_host::Ristretto255Traits>("Hello MPVSS Example (Ristretto255).", argc, argv);
}
