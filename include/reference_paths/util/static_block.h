// Forwarding header: lets sources written against the reference's include path "util/static_block.h" compile unchanged.
#pragma once
#include "giddy/util/static_block.hpp"
