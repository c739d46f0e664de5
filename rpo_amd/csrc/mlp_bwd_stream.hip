// The large-batch streaming backward (mlp_bwd_stream.h: rows kernel + weights kernel + TD prologue) as a translation unit of its
// own, compiled WITHOUT SLP vectorisation (rpo_amd/csrc/build.py FILE_FLAGS).  Under plain -O3 the SLP pass packs the per-lane
// partial sums (db0, dW1, the dh formation) into v_pk_* instructions behind register moves; packed f32 vector work beside f32
// MFMAs is an anti-lever (they share the vector lanes): rocprof-free A/B at 2^20 rows, two alternating rounds on one box:
// backward with every parameter gradient 1 179 -> 1 144 us (profiles/r06_ab_noslp.txt).  The forward kernels measured 1 % slower
// with the flag and keep SLP, hence the separate unit.
#include <stdlib.h>

#include "heads_dev.h"
#include "mlp_bwd.h"
#include "mlp_tile.h"
#include "mlp_bwd_stream.h"

namespace rpo_mlp_dev {

bool bwd_stream_applies_x(const BwdArgs& a, const SplitK& k) { return bwd_stream_applies(a, k); }
int launch_bwd_stream_x(const BwdArgs& a, const SplitK& k, hipStream_t stream) { return launch_bwd_stream(a, k, stream); }

}  // namespace rpo_mlp_dev
