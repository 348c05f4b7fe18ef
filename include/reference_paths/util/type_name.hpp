// Forwarding header: lets sources written against the reference's include path "util/type_name.hpp" compile unchanged.
#pragma once
#include "giddy/util/type_name.hpp"
