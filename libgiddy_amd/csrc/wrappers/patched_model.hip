// Instantiates and registers the fused Model + patches wrappers (an extension beyond the reference's 163
// registrations): the tuples of the plain Model wrappers, keys "decompression::model::patched::kernel_t<...>".
#include "giddy/kernel_wrappers/decompression/patched/model.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace model {
namespace patched {

static_block {
	namespace unary_models = ::cuda::functors::unary::parametric_model;

	//         IndexSize  Uncompressed   UnaryModelFunction
	kernel_t < 4,         int16_t,       unary_models::linear  < 4, int16_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       unary_models::linear  < 4, int32_t >  >::registerInSubclassFactory();
	kernel_t < 8,         int32_t,       unary_models::linear  < 8, int32_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int16_t,       unary_models::constant< 4, int16_t >  >::registerInSubclassFactory();
	kernel_t < 4,         int32_t,       unary_models::constant< 4, int32_t >  >::registerInSubclassFactory();
}

} // namespace patched
} // namespace model
} // namespace decompression
} // namespace kernels
} // namespace cuda
