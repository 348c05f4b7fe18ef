// Forwarding header: lets sources written against the reference's include path "kernels/decompression/frame_of_reference.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/frame_of_reference.hpp"
