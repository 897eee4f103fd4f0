#pragma once
#include "../core/core.hpp"   // TEST INFRASTRUCTURE ONLY (see core/core.hpp)
