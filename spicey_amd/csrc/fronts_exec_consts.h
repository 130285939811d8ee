// constants of the dense-front engine shared by device code (fronts_exec.h) and host code
#pragma once
#define SPICEY_FB 16     // panel width = MFMA tile edge
#define SPICEY_LPLD 17   // row stride (doubles) of a staged L panel in LDS: odd, so that a thread-per-row walk is bank-conflict free
#define SPICEY_FRONT_LDS_DOUBLES 19456  // 152 KiB of LDS scratch per workgroup when a program has fronts
#define SPICEY_FRONT_LDS_PAD 17  // an LDS-resident front has row stride Mp + 17 (odd; room for the right-hand-side tile)
#ifndef SPICEY_TRAIL_TILES_STAGED
#define SPICEY_TRAIL_TILES_STAGED 2  // 16 x 16 tiles a wave keeps in flight per turn of the right-looking update of a staged front (the fallback path; all operands of a turn are loaded before its first MFMA: 24 registers per tile)
#endif
