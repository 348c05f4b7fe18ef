// Forwarding header: lets sources written against the reference's include path "cuda/kernel_wrapper.cuh" compile unchanged.
#pragma once
#include "giddy/kernel_wrapper.hpp"
