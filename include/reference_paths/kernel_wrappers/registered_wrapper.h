// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/registered_wrapper.h" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/registered_wrapper.hpp"
