// Instantiates and registers the Model wrappers — the same 5 tuples as the reference's
// src/kernel_wrappers/decompression/model.cu:92-96.
#include "giddy/kernel_wrappers/decompression/model.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace model {

static_block {
	namespace unary_models = ::cuda::functors::unary::parametric_model;

	//         IndexSize  Uncompressed   UnaryModelFunction
	kernel_t < 4,         int16_t,       unary_models::linear  < 4, int16_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 8,         int32_t,       unary_models::linear  < 8, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int16_t,       unary_models::constant< 4, int16_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       unary_models::constant< 4, int32_t >  >::registerInSubclassFactory();
}

} // namespace model
} // namespace decompression
} // namespace kernels
} // namespace cuda
