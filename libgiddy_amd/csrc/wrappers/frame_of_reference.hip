// Instantiates and registers the Frame-of-Reference wrappers — the same 8 tuples as the reference's
// src/kernel_wrappers/decompression/frame_of_reference.cu:132-142.
#include "giddy/kernel_wrappers/decompression/frame_of_reference.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace frame_of_reference {

static_block {
	namespace unary_models = ::cuda::functors::unary::parametric_model;

	//         IndexSize  Uncompressed   Compressed   UnaryModelFunction
	kernel_t < 4,         int16_t,       int8_t,      unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       int8_t,      unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       int16_t,     unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 8,         int32_t,       int16_t,     unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 8,         int32_t,       int16_t,     unary_models::linear  < 8, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int16_t,       int8_t,      unary_models::constant< 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       int8_t,      unary_models::constant< 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       int16_t,     unary_models::constant< 4, int32_t >  >::registerInSubclassFactory();
}

} // namespace frame_of_reference
} // namespace decompression
} // namespace kernels
} // namespace cuda
