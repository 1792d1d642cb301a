// Counterpart of the reference's examples/mpvss_all_secp256k1.rs:10-92 over the C++ mirror (see mpvss_all_ec.hpp).
//   build: make -C mpvss_rs_amd/csrc examples      run: ./examples/mpvss_all_secp256k1 [seed]
#include "mpvss_all_ec.hpp"

int main(int argc, char** argv) {
  return run_mpvss_all<mpvss_host::Secp256k1Traits>("Hello MPVSS Example (secp256k1).", argc, argv);
}
