#pragma once
#include "Frame.h"   // TEST INFRASTRUCTURE ONLY (see Frame.h)
