// Forwarding header: lets sources written against the reference's include path "kernels/decompression/discard_zero_bytes_variable.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/discard_zero_bytes_variable.hpp"
