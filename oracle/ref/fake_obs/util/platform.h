/* fake libobs header (test infrastructure): os_gettime_ns lives in obs-module.h */
#pragma once
#include "../obs-module.h"
