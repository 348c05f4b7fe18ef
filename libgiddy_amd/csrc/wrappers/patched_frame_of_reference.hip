// Instantiates and registers the fused Frame-of-Reference + patches wrappers (an extension beyond the reference's
// 163 registrations): the tuples of the plain FOR wrappers, keys "decompression::frame_of_reference::patched::kernel_t<...>".
#include "giddy/kernel_wrappers/decompression/patched/frame_of_reference.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace frame_of_reference {
namespace patched {

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

} // namespace patched
} // namespace frame_of_reference
} // namespace decompression
} // namespace kernels
} // namespace cuda
