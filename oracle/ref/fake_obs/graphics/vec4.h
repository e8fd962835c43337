/* fake libobs header (test infrastructure): vec4 lives in obs-module.h */
#pragma once
#include "../obs-module.h"
