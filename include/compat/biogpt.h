/* forwarding header: lets code written against the reference include "biogpt.h" unchanged */
#pragma once
#include "../biogpt_compat.h"
