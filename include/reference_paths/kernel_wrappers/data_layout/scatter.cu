// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/data_layout/scatter.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/data_layout/scatter.hpp"
