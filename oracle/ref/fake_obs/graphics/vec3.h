/* fake libobs header (test infrastructure): vec3 lives in obs-module.h */
#pragma once
#include "../obs-module.h"
