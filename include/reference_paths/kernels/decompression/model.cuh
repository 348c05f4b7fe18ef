// Forwarding header: lets sources written against the reference's include path "kernels/decompression/model.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/model.hpp"
