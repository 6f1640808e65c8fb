// forwarding header: diffusion_step live in core.hpp (kept so the reference's include names still work)
#pragma once
#include "core.hpp"
