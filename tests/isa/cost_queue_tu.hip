// Instantiates the cost kernels that keep a hand-counted memory queue (k_cost2, gray cost, D <= 128) so that
// tests/test_cost_queue_isa.py can read their gfx950 listing without compiling the whole library.
#include "../../stereo-vision_amd/csrc/sv_kernels.hip.h"
#define INST(N, NT) template __global__ void svk::k_cost2<N, NT, 1>(const uint2*, const uint16_t*, uint32_t*, int, int, int, int, int, uint32_t, int, int, int, int, int, int, int, int, int, int, uint32_t*);
INST(1, 512) INST(2, 512) INST(3, 512) INST(4, 512)
INST(1, 1024) INST(2, 1024) INST(3, 1024) INST(4, 1024)
