// stand-in for the reference's lib/constants/EPS.ts on the GPU box (where /root/reference does not exist)
export const EPS = 1e-15
