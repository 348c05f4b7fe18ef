// Forwarding header: lets sources written against the reference's include path "kernels/decompression/run_position_encoding.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/run_position_encoding.hpp"
