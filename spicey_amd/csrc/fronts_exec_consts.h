// constants of the dense-front engine shared by device code (fronts_exec.h) and host code
#pragma once
#define SPICEY_FB 16     // panel width = MFMA tile edge
#define SPICEY_LPLD 17   // row stride (doubles) of a staged L panel in LDS: odd, so that a thread-per-row walk is bank-conflict free
#define SPICEY_FRONT_LDS_DOUBLES 19456  // 152 KiB of LDS scratch per workgroup when a program has fronts
#define SPICEY_FRONT_LDS_PAD 17  // an LDS-resident front has row stride Mp + 17 (odd; room for the right-hand-side tile)
