import numpy as np
# least-squares minimax-ish fit on Chebyshev nodes of [0,(pi/4)^2] for
#   sin(r) = r + r*z*(S1 + z*S2 + z*z*S3),  cos(r) = 1 + z*(-0.5 + z*(C2 + z*C3 + z*z*C4)),  z=r*r
a=np.pi/4*1.02
N=4000
k=np.arange(N)
r=a*np.cos(np.pi*(k+0.5)/(2*N))   # chebyshev-ish nodes in (0,a]
z=r*r
ys=(np.sin(r)/r-1)/z
A=np.stack([np.ones_like(z),z,z*z],1)
# weight for relative error of sin: error = r*z*dP  -> relative = z*dP/(sin r/r) ~ z*dP
W=z
S=np.linalg.lstsq(A*W[:,None],ys*W,rcond=None)[0]
yc=((np.cos(r)-1)/z+0.5)/z
B=np.stack([np.ones_like(z),z,z*z],1)
W2=z*z
Cc=np.linalg.lstsq(B*W2[:,None],yc*W2,rcond=None)[0]
print('S',[float(np.float32(v)) for v in S],[hex(np.float32(v).view(np.uint32)) for v in S])
print('C',[float(np.float32(v)) for v in Cc],[hex(np.float32(v).view(np.uint32)) for v in Cc])
