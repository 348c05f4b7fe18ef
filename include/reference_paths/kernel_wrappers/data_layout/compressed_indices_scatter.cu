// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/data_layout/compressed_indices_scatter.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/data_layout/compressed_indices_scatter.hpp"
