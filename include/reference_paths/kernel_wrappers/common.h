// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/common.h" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
