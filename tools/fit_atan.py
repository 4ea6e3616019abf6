import numpy as np, sys
LD=np.longdouble
Z=LD(np.tan(np.pi/8))**2*LD(1.0005)   # slightly beyond: red threshold 70/169 is just below tan(pi/8)... r<=tan(pi/8)+eps
# after reduction r <= max(70/169, (1-70/169)/(1+70/169)) 
t=LD(70)/LD(169); rmax=max(t,(1-t)/(1+t)); Z=rmax*rmax*LD(1.0000001)
def f(z):
    z=np.asarray(z,dtype=LD); s=np.zeros_like(z); 
    for k in range(80,0,-1):
        s=s*z+LD((-1)**k)/LD(2*k+1)
    return s   # = sum_{k>=1} (-1)^k z^(k-1)/(2k+1): atan(r) = r + r*z*f(z)
def cheb_fit(n):
    k=np.arange(n+1,dtype=LD)
    x=np.cos(LD(np.pi)*(2*k+1)/(2*(n+1)))  # nodes in [-1,1] (double-precision pi: fine)
    z=(x+1)*Z/2
    V=np.vander(z,n+1,increasing=True).astype(LD)
    # solve in long double via numpy? use float128-unfriendly linalg: do Gaussian elimination manually
    A=V.copy(); b=f(z).copy(); m=n+1
    for i in range(m):
        p=i+np.argmax(np.abs(A[i:,i])); A[[i,p]]=A[[p,i]]; b[[i,p]]=b[[p,i]]
        for j in range(i+1,m):
            fct=A[j,i]/A[i,i]; A[j,i:]-=fct*A[i,i:]; b[j]-=fct*b[i]
    c=np.zeros(m,dtype=LD)
    for i in range(m-1,-1,-1):
        c[i]=(b[i]-np.dot(A[i,i+1:],c[i+1:]))/A[i,i]
    return c
for n in range(5,11):
    c=cheb_fit(n); cd=c.astype(np.float64)
    zz=np.linspace(0,float(Z),200001).astype(LD)
    p=np.zeros_like(zz)
    for ci in cd[::-1]: p=p*zz+LD(ci)
    err=np.max(np.abs((p-f(zz))*zz))   # relative error of atan(r)/r  ~ z*|dQ|
    print(n, float(err), " ".join(float(v).hex() for v in cd[::-1]))
