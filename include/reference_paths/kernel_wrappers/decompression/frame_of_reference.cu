// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/decompression/frame_of_reference.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/decompression/frame_of_reference.hpp"
