// ORACLE — TEST INFRASTRUCTURE ONLY (see orc_math.h header).
// BSDFs, materials and lights of the bundled scenes:
// core/reflection.{h,cpp}, core/microfacet.{h,cpp}, materials/{matte,plastic}.cpp,
// lights/{point,distant,diffuse}.cpp, shapes/sphere.cpp (sampling), core/sampling.{h,cpp}.
#pragma once
#include "orc_accel.h"

namespace orc {

// core/sampling.cpp:113-130
inline P2 ConcentricSampleDisk(const P2 &u) {
    P2 uOffset(2.f * u.x - 1, 2.f * u.y - 1);
    if (uOffset.x == 0 && uOffset.y == 0) return P2(0, 0);
    Float theta, r;
    if (std::abs(uOffset.x) > std::abs(uOffset.y)) {
        r = uOffset.x;
        theta = PiOver4 * (uOffset.y / uOffset.x);
    } else {
        r = uOffset.y;
        theta = PiOver2 - PiOver4 * (uOffset.x / uOffset.y);
    }
    return P2(r * m_cosf(theta), r * m_sinf(theta));
}
// core/sampling.h:159-163
inline V3 CosineSampleHemisphere(const P2 &u) {
    P2 d = ConcentricSampleDisk(u);
    Float z = std::sqrt(smax((Float)0, 1 - d.x * d.x - d.y * d.y));
    return V3(d.x, d.y, z);
}
// core/sampling.cpp:98-103
inline V3 UniformSampleSphere(const P2 &u) {
    Float z = 1 - 2 * u.x;
    Float r = std::sqrt(smax((Float)0, (Float)1 - z * z));
    Float phi = 2 * Pi * u.y;
    return V3(r * m_cosf(phi), r * m_sinf(phi), z);
}
inline Float UniformConePdf(Float cosThetaMax) { return 1 / (2 * Pi * (1 - cosThetaMax)); }
// core/sampling.h:171-174
inline Float PowerHeuristic(int nf, Float fPdf, int ng, Float gPdf) {
    Float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}

// core/reflection.h:56-122
inline Float CosTheta(const V3 &w) { return w.z; }
inline Float Cos2Theta(const V3 &w) { return w.z * w.z; }
inline Float AbsCosTheta(const V3 &w) { return std::abs(w.z); }
inline Float Sin2Theta(const V3 &w) { return smax((Float)0, (Float)1 - Cos2Theta(w)); }
inline Float SinTheta(const V3 &w) { return std::sqrt(Sin2Theta(w)); }
inline Float TanTheta(const V3 &w) { return SinTheta(w) / CosTheta(w); }
inline Float Tan2Theta(const V3 &w) { return Sin2Theta(w) / Cos2Theta(w); }
inline Float CosPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 1 : Clamp(w.x / s, -1, 1); }
inline Float SinPhi(const V3 &w) { Float s = SinTheta(w); return (s == 0) ? 0 : Clamp(w.y / s, -1, 1); }
inline Float Cos2Phi(const V3 &w) { return CosPhi(w) * CosPhi(w); }
inline Float Sin2Phi(const V3 &w) { return SinPhi(w) * SinPhi(w); }
inline V3 Reflect(const V3 &wo, const V3 &n) { return -wo + 2 * Dot(wo, n) * n; }
inline bool SameHemisphere(const V3 &w, const V3 &wp) { return w.z * wp.z > 0; }
// core/reflection.h:96-108
inline bool Refract(const V3 &wi, const V3 &n, Float eta, V3 *wt) {
    Float cosThetaI = Dot(n, wi);
    Float sin2ThetaI = smax(Float(0), Float(1 - cosThetaI * cosThetaI));
    Float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    Float cosThetaT = std::sqrt(1 - sin2ThetaT);
    *wt = eta * -wi + (eta * cosThetaI - cosThetaT) * n;
    return true;
}

// core/reflection.cpp:47-68
inline Float FrDielectric(Float cosThetaI, Float etaI, Float etaT) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    bool entering = cosThetaI > 0.f;
    if (!entering) { std::swap(etaI, etaT); cosThetaI = std::abs(cosThetaI); }
    Float sinThetaI = std::sqrt(smax((Float)0, 1 - cosThetaI * cosThetaI));
    Float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    Float cosThetaT = std::sqrt(smax((Float)0, 1 - sinThetaT * sinThetaT));
    Float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    Float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}

// core/microfacet.h:123-128.  std::log(float) is glibc logf in the reference;
// constant textures make this a per-material constant, evaluated on the host.
inline Float RoughnessToAlpha(Float roughness) {
    roughness = smax(roughness, (Float)1e-3);
    Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

// TrowbridgeReitzDistribution, sampleVisibleArea = true (core/microfacet.cpp)
struct TRDist {
    Float alphax, alphay;
    Float D(const V3 &wh) const {                 // :163-171
        Float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        const Float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        Float e = (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay)) * tan2Theta;
        return 1 / (Pi * alphax * alphay * cos4Theta * (1 + e) * (1 + e));
    }
    Float Lambda(const V3 &w) const {             // :185-193
        Float absTanTheta = std::abs(TanTheta(w));
        if (std::isinf(absTanTheta)) return 0.;
        Float alpha = std::sqrt(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        Float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
        return (-1 + std::sqrt(1.f + alpha2Tan2Theta)) / 2;
    }
    Float G1(const V3 &w) const { return 1 / (1 + Lambda(w)); }                  // microfacet.h:54-57
    Float G(const V3 &wo, const V3 &wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    Float Pdf(const V3 &wo, const V3 &wh) const {  // :338-344
        return D(wh) * G1(wo) * AbsDot(wo, wh) / AbsCosTheta(wo);
    }
    // :238-280
    static void Sample11(Float cosTheta, Float U1, Float U2, Float *slope_x, Float *slope_y) {
        if (cosTheta > .9999) {
            Float r = std::sqrt(U1 / (1 - U1));
            Float phi = 6.28318530718 * U2;
            *slope_x = r * m_cos(phi);   // float * double -> double -> Float
            *slope_y = r * m_sin(phi);
            return;
        }
        Float sinTheta = std::sqrt(smax((Float)0, (Float)1 - cosTheta * cosTheta));
        Float tanTheta = sinTheta / cosTheta;
        Float a = 1 / tanTheta;
        Float G1 = 2 / (1 + std::sqrt(1.f + 1.f / (a * a)));
        Float A = 2 * U1 / G1 - 1;
        Float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10) tmp = 1e10;
        Float B = tanTheta;
        Float D = std::sqrt(smax(Float(B * B * tmp * tmp - (A * A - B * B) * tmp), Float(0)));
        Float slope_x_1 = B * tmp - D;
        Float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        Float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        Float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * std::sqrt(1.f + *slope_x * *slope_x);
    }
    // :282-305
    static V3 Sample(const V3 &wi, Float alpha_x, Float alpha_y, Float U1, Float U2) {
        V3 wiStretched = Normalize(V3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
        Float slope_x, slope_y;
        Sample11(CosTheta(wiStretched), U1, U2, &slope_x, &slope_y);
        Float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alpha_x * slope_x;
        slope_y = alpha_y * slope_y;
        return Normalize(V3(-slope_x, -slope_y, 1.));
    }
    V3 Sample_wh(const V3 &wo, const P2 &u) const {   // :307-336, visible-area branch
        bool flip = wo.z < 0;
        V3 wh = Sample(flip ? -wo : wo, alphax, alphay, u.x, u.y);
        if (flip) wh = -wh;
        return wh;
    }
};

enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16,
       BSDF_ALL = 31 };
enum { BXDF_LAMBERT = 0, BXDF_MICROFACET = 1, BXDF_SPECULAR_REFLECTION = 2, BXDF_OREN_NAYAR = 3, BXDF_FRESNEL_BLEND = 4,
       BXDF_MICROFACET_CONDUCTOR = 5, BXDF_FRESNEL_SPECULAR = 6, BXDF_SPECULAR_TRANSMISSION = 7, BXDF_MICROFACET_TRANSMISSION = 8 };

inline Spec SqrtS(const Spec &s) { return Spec(std::sqrt(s.c[0]), std::sqrt(s.c[1]), std::sqrt(s.c[2])); }
// FrConductor, core/reflection.cpp:70-95
inline Spec FrConductor(Float cosThetaI, const Spec &etai, const Spec &etat, const Spec &k) {
    cosThetaI = Clamp(cosThetaI, -1, 1);
    Spec eta = etat / etai;
    Spec etak = k / etai;
    Float cosThetaI2 = cosThetaI * cosThetaI;
    Float sinThetaI2 = (Float)(1. - (double)cosThetaI2);
    Spec eta2 = eta * eta;
    Spec etak2 = etak * etak;
    Spec t0 = eta2 - etak2 - Spec(sinThetaI2);
    Spec a2plusb2 = SqrtS(t0 * t0 + 4 * eta2 * etak2);
    Spec t1 = a2plusb2 + Spec(cosThetaI2);
    Spec a = SqrtS(0.5f * (a2plusb2 + t0));
    Spec t2 = (Float)2 * cosThetaI * a;
    Spec Rs = (t1 - t2) / (t1 + t2);
    Spec t3 = cosThetaI2 * a2plusb2 + Spec(sinThetaI2 * sinThetaI2);
    Spec t4 = t2 * sinThetaI2;
    Spec Rp = Rs * (t3 - t4) / (t3 + t4);
    return (Float)0.5 * (Rp + Rs);
}

struct BxDF {
    int kind; int type;
    Spec R;
    TRDist dist;   // microfacet only; Fresnel is FresnelDielectric(1.5, 1) (plastic.cpp:56)
    Float A = 1, B = 0;   // OrenNayar (core/reflection.h:414-420)
    Spec S, K;            // FresnelBlend: S = Rs; conductor microfacet: S = eta, K = k; FresnelSpecular: S = T
    Float etaA = 1, etaB = 1;      // FresnelSpecular, SpecularTransmission (TransportMode::Radiance)
    // Fresnel term of SpecularReflection / MicrofacetReflection when it is not the kind's usual one (mirror: FresnelNoOp; plastic:
    // FresnelDielectric(1.5, 1)): UberMaterial's lobes carry FresnelDielectric(1, e) (materials/uber.cpp:73, 95)
    bool frDielectric = false; Float frEtaI = 1, frEtaT = 1;
    bool MatchesFlags(int t) const { return (type & t) == type; }
    Spec f(const V3 &wo, const V3 &wi) const {
        if (kind == BXDF_LAMBERT) return R * InvPi;       // reflection.cpp:178-180
        if (kind == BXDF_SPECULAR_REFLECTION || kind == BXDF_FRESNEL_SPECULAR || kind == BXDF_SPECULAR_TRANSMISSION) return Spec(0.f);      // reflection.h:199-201, 337-339, 527-529
        if (kind == BXDF_FRESNEL_BLEND) {                 // reflection.cpp:285-298; R = Rd, S = Rs
            auto pow5 = [](Float v) { return (v * v) * (v * v) * v; };
            Spec diffuse = (28.f / (23.f * Pi)) * R * (Spec(1.f) - S) * (1 - pow5(1 - .5f * AbsCosTheta(wi))) * (1 - pow5(1 - .5f * AbsCosTheta(wo)));
            V3 wh = wi + wo;
            if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0);
            wh = Normalize(wh);
            Spec schlick = S + pow5(1 - Dot(wi, wh)) * (Spec(1.) - S);      // SchlickFresnel, reflection.h:485-488
            Spec specular = dist.D(wh) / (4 * AbsDot(wi, wh) * smax(AbsCosTheta(wi), AbsCosTheta(wo))) * schlick;
            return diffuse + specular;
        }
        if (kind == BXDF_OREN_NAYAR) {                    // reflection.cpp:197-219
            Float sinThetaI = SinTheta(wi), sinThetaO = SinTheta(wo);
            Float maxCos = 0;
            if (sinThetaI > 1e-4 && sinThetaO > 1e-4) {
                Float sinPhiI = SinPhi(wi), cosPhiI = CosPhi(wi);
                Float sinPhiO = SinPhi(wo), cosPhiO = CosPhi(wo);
                Float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
                maxCos = smax((Float)0, dCos);
            }
            Float sinAlpha, tanBeta;
            if (AbsCosTheta(wi) > AbsCosTheta(wo)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / AbsCosTheta(wi); }
            else { sinAlpha = sinThetaI; tanBeta = sinThetaO / AbsCosTheta(wo); }
            return R * InvPi * (A + B * maxCos * sinAlpha * tanBeta);
        }
        if (kind == BXDF_MICROFACET_TRANSMISSION) {       // MicrofacetTransmission::f, reflection.cpp:244-266 (R = T; FresnelDielectric(etaA, etaB); TransportMode::Radiance)
            if (SameHemisphere(wo, wi)) return 0;      // transmission only
            Float cosThetaO = CosTheta(wo);
            Float cosThetaI = CosTheta(wi);
            if (cosThetaI == 0 || cosThetaO == 0) return Spec(0);
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            if (wh.z < 0) wh = -wh;
            Spec F = Spec(FrDielectric(Dot(wo, wh), etaA, etaB));
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float factor = 1 / eta;
            return (Spec(1.f) - F) * R *
                   std::abs(dist.D(wh) * dist.G(wo, wi) * eta * eta * AbsDot(wi, wh) * AbsDot(wo, wh) * factor * factor /
                            (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
        }
        // reflection.cpp:226-236
        Float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return Spec(0.);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return Spec(0.);
        wh = Normalize(wh);
        // FresnelDielectric(1.5, 1) (plastic) or FresnelConductor(1, eta = S, k = K) (metal: Evaluate takes |cos|, reflection.cpp:118-120)
        Spec F = kind == BXDF_MICROFACET_CONDUCTOR ? FrConductor(std::abs(Dot(wi, wh)), Spec(1.), S, K)
                                                   : Spec(frDielectric ? FrDielectric(Dot(wi, wh), frEtaI, frEtaT) : FrDielectric(Dot(wi, wh), 1.5f, 1.f));
        return R * dist.D(wh) * dist.G(wo, wi) * F / (4 * cosThetaI * cosThetaO);
    }
    Float Pdf(const V3 &wo, const V3 &wi) const {
        if (kind == BXDF_SPECULAR_REFLECTION || kind == BXDF_FRESNEL_SPECULAR || kind == BXDF_SPECULAR_TRANSMISSION) return 0;               // reflection.h:204, 342, 532
        if (kind == BXDF_LAMBERT || kind == BXDF_OREN_NAYAR) return SameHemisphere(wo, wi) ? AbsCosTheta(wi) * InvPi : 0;   // :387-389
        if (kind == BXDF_FRESNEL_BLEND) {                                                        // :470-475
            if (!SameHemisphere(wo, wi)) return 0;
            V3 wh = Normalize(wo + wi);
            Float pdf_wh = dist.Pdf(wo, wh);
            return .5f * (AbsCosTheta(wi) * InvPi + pdf_wh / (4 * Dot(wo, wh)));
        }
        if (kind == BXDF_MICROFACET_TRANSMISSION) {                                              // :437-447
            if (SameHemisphere(wo, wi)) return 0;
            Float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            V3 wh = Normalize(wo + wi * eta);
            Float sqrtDenom = Dot(wo, wh) + eta * Dot(wi, wh);
            Float dwh_dwi = std::abs((eta * eta * Dot(wi, wh)) / (sqrtDenom * sqrtDenom));
            return dist.Pdf(wo, wh) * dwh_dwi;
        }
        if (!SameHemisphere(wo, wi)) return 0;                                                   // :416-420
        V3 wh = Normalize(wo + wi);
        return dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
    }
    Spec Sample_f(const V3 &wo, V3 *wi, const P2 &u, Float *pdf, int *sampledType = nullptr) const {
        if (kind == BXDF_FRESNEL_SPECULAR) {               // reflection.cpp:477-512
            Float F = FrDielectric(CosTheta(wo), etaA, etaB);
            if (u.x < F) {
                *wi = V3(-wo.x, -wo.y, wo.z);
                if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_REFLECTION;
                *pdf = F;
                return F * R / AbsCosTheta(*wi);
            } else {
                bool entering = CosTheta(wo) > 0;
                Float etaI = entering ? etaA : etaB;
                Float etaT = entering ? etaB : etaA;
                if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return 0;
                Spec ft = S * (1 - F);
                ft *= (etaI * etaI) / (etaT * etaT);
                if (sampledType) *sampledType = BSDF_SPECULAR | BSDF_TRANSMISSION;
                *pdf = 1 - F;
                return ft / AbsCosTheta(*wi);
            }
        }
        if (kind == BXDF_SPECULAR_REFLECTION) {            // :136-143, FresnelNoOp::Evaluate == Spectrum(1.)
            *wi = V3(-wo.x, -wo.y, wo.z);
            *pdf = 1;
            return (frDielectric ? Spec(FrDielectric(CosTheta(*wi), frEtaI, frEtaT)) : Spec(1.f)) * R / AbsCosTheta(*wi);
        }
        if (kind == BXDF_SPECULAR_TRANSMISSION) {          // :145-163 (R = T; fresnel = FresnelDielectric(etaA, etaB); TransportMode::Radiance)
            bool entering = CosTheta(wo) > 0;
            Float etaI = entering ? etaA : etaB;
            Float etaT = entering ? etaB : etaA;
            if (!Refract(wo, Faceforward(V3(0, 0, 1), wo), etaI / etaT, wi)) return 0;
            *pdf = 1;
            Spec ft = R * (Spec(1.) - Spec(FrDielectric(CosTheta(*wi), etaA, etaB)));
            ft *= (etaI * etaI) / (etaT * etaT);
            return ft / AbsCosTheta(*wi);
        }
        if (kind == BXDF_FRESNEL_BLEND) {                  // :450-468
            P2 uu = u;
            if (uu.x < .5) {
                uu.x = smin(2 * uu.x, OneMinusEpsilon);
                *wi = CosineSampleHemisphere(uu);
                if (wo.z < 0) wi->z *= -1;
            } else {
                uu.x = smin(2 * (uu.x - .5f), OneMinusEpsilon);
                V3 wh = dist.Sample_wh(wo, uu);
                *wi = Reflect(wo, wh);
                if (!SameHemisphere(wo, *wi)) return Spec(0.f);
            }
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        if (kind == BXDF_LAMBERT || kind == BXDF_OREN_NAYAR) {      // :378-385
            *wi = CosineSampleHemisphere(u);
            if (wo.z < 0) wi->z *= -1;
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        if (kind == BXDF_MICROFACET_TRANSMISSION) {        // :425-435
            if (wo.z == 0) return 0.;
            V3 wh = dist.Sample_wh(wo, u);
            Float eta = CosTheta(wo) > 0 ? (etaA / etaB) : (etaB / etaA);
            if (!Refract(wo, wh, eta, wi)) return 0;
            *pdf = Pdf(wo, *wi);
            return f(wo, *wi);
        }
        // :402-414
        if (wo.z == 0) return 0.;
        V3 wh = dist.Sample_wh(wo, u);
        *wi = Reflect(wo, wh);
        if (!SameHemisphere(wo, *wi)) return Spec(0.f);
        *pdf = dist.Pdf(wo, wh) / (4 * Dot(wo, wh));
        return f(wo, *wi);
    }
};

// core/reflection.h:153-202, reflection.cpp:670-785
struct BSDF {
    V3 ns, ng, ss, ts;
    int nBxDFs = 0;
    BxDF bxdfs[5];      // UberMaterial adds up to five (materials/uber.cpp)
    Float eta = 1;        // BSDF::eta (core/reflection.h:156-157)
    void Init(const SurfaceInteraction &si) {
        eta = 1;
        ns = si.shading.n; ng = si.n;
        ss = Normalize(si.shading.dpdu);
        ts = Cross(ns, ss);
        nBxDFs = 0;
    }
    int NumComponents(int flags) const {
        int num = 0;
        for (int i = 0; i < nBxDFs; ++i) if (bxdfs[i].MatchesFlags(flags)) ++num;
        return num;
    }
    V3 WorldToLocal(const V3 &v) const { return V3(Dot(v, ss), Dot(v, ts), Dot(v, ns)); }
    V3 LocalToWorld(const V3 &v) const {
        return V3(ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                  ss.z * v.x + ts.z * v.y + ns.z * v.z);
    }
    Spec f(const V3 &woW, const V3 &wiW, int flags) const {
        V3 wi = WorldToLocal(wiW), wo = WorldToLocal(woW);
        if (wo.z == 0) return 0.;
        bool reflect = Dot(wiW, ng) * Dot(woW, ng) > 0;
        Spec f(0.f);
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags) &&
                ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) || (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
                f += bxdfs[i].f(wo, wi);
        return f;
    }
    Float Pdf(const V3 &woWorld, const V3 &wiWorld, int flags) const {
        if (nBxDFs == 0.f) return 0.f;
        V3 wo = WorldToLocal(woWorld), wi = WorldToLocal(wiWorld);
        if (wo.z == 0) return 0.;
        Float pdf = 0.f;
        int matchingComps = 0;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(flags)) { ++matchingComps; pdf += bxdfs[i].Pdf(wo, wi); }
        Float v = matchingComps > 0 ? pdf / matchingComps : 0.f;
        return v;
    }
    // *pdf is deliberately left untouched on the early "wo.z == 0" return, as
    // in the reference (:729-730); callers initialise it as the reference does.
    Spec Sample_f(const V3 &woWorld, V3 *wiWorld, const P2 &u, Float *pdf, int type, int *sampledType) const {
        int matchingComps = NumComponents(type);
        if (matchingComps == 0) { *pdf = 0; if (sampledType) *sampledType = 0; return Spec(0); }
        int comp = smin((int)std::floor(u.x * matchingComps), matchingComps - 1);
        const BxDF *bxdf = nullptr;
        int count = comp;
        for (int i = 0; i < nBxDFs; ++i)
            if (bxdfs[i].MatchesFlags(type) && count-- == 0) { bxdf = &bxdfs[i]; break; }
        P2 uRemapped(smin(u.x * matchingComps - comp, OneMinusEpsilon), u.y);
        V3 wi, wo = WorldToLocal(woWorld);
        if (wo.z == 0) return 0.;
        *pdf = 0;
        if (sampledType) *sampledType = bxdf->type;
        Spec f = bxdf->Sample_f(wo, &wi, uRemapped, pdf, sampledType);
        if (*pdf == 0) { if (sampledType) *sampledType = 0; return 0; }
        *wiWorld = LocalToWorld(wi);
        if (!(bxdf->type & BSDF_SPECULAR) && matchingComps > 1)
            for (int i = 0; i < nBxDFs; ++i)
                if (&bxdfs[i] != bxdf && bxdfs[i].MatchesFlags(type)) *pdf += bxdfs[i].Pdf(wo, wi);
        if (matchingComps > 1) *pdf /= matchingComps;
        if (!(bxdf->type & BSDF_SPECULAR)) {
            bool reflect = Dot(*wiWorld, ng) * Dot(woWorld, ng) > 0;
            f = 0.;
            for (int i = 0; i < nBxDFs; ++i)
                if (bxdfs[i].MatchesFlags(type) &&
                    ((reflect && (bxdfs[i].type & BSDF_REFLECTION)) ||
                     (!reflect && (bxdfs[i].type & BSDF_TRANSMISSION))))
                    f += bxdfs[i].f(wo, wi);
        }
        return f;
    }
};

// ---- image textures: MIPMap lookups (core/mipmap.h:203-338) and UVMapping2D (core/texture.cpp:93-99) ----
inline int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }
inline Spec MipTexel(const Texture &tx, int level, int s, int t) {
    const MipLevel &l = tx.levels[level];
    if (tx.wrap == WRAP_REPEAT) { s = ModI(s, l.w); t = ModI(t, l.h); }
    else if (tx.wrap == WRAP_CLAMP) { s = Clamp(s, 0, l.w - 1); t = Clamp(t, 0, l.h - 1); }
    else if (s < 0 || s >= l.w || t < 0 || t >= l.h) return Spec(0.f);
    const Float *p = &l.rgb[3 * ((size_t)t * l.w + s)];
    return Spec(p[0], p[1], p[2]);
}
inline Spec MipTriangle(const Texture &tx, int level, const P2 &st) {
    const int nLevels = (int)tx.levels.size();
    level = Clamp(level, 0, nLevels - 1);
    Float s = st.x * tx.levels[level].w - 0.5f;
    Float t = st.y * tx.levels[level].h - 0.5f;
    int s0 = (int)std::floor(s), t0 = (int)std::floor(t);
    Float ds = s - s0, dt = t - t0;
    return (1 - ds) * (1 - dt) * MipTexel(tx, level, s0, t0) + (1 - ds) * dt * MipTexel(tx, level, s0, t0 + 1) +
           ds * (1 - dt) * MipTexel(tx, level, s0 + 1, t0) + ds * dt * MipTexel(tx, level, s0 + 1, t0 + 1);
}
inline Float Log2F(Float x) { const Float invLog2 = 1.442695040888963387004650940071; return m_logf(x) * invLog2; }   // core/pbrt.h:328-331
inline Spec LerpS(Float t, const Spec &a, const Spec &b) { return (1 - t) * a + t * b; }
inline Spec MipLookupWidth(const Texture &tx, const P2 &st, Float width) {
    const int nLevels = (int)tx.levels.size();
    Float level = nLevels - 1 + Log2F(smax(width, (Float)1e-8));
    if (level < 0) return MipTriangle(tx, 0, st);
    else if (level >= nLevels - 1) return MipTexel(tx, nLevels - 1, 0, 0);
    int iLevel = (int)std::floor(level);
    Float delta = level - iLevel;
    return LerpS(delta, MipTriangle(tx, iLevel, st), MipTriangle(tx, iLevel + 1, st));
}
inline Spec MipEWA(const Texture &tx, int level, P2 st, P2 dst0, P2 dst1) {
    const int nLevels = (int)tx.levels.size();
    if (level >= nLevels) return MipTexel(tx, nLevels - 1, 0, 0);
    const MipLevel &l = tx.levels[level];
    st.x = st.x * l.w - 0.5f; st.y = st.y * l.h - 0.5f;
    dst0.x *= l.w; dst0.y *= l.h; dst1.x *= l.w; dst1.y *= l.h;
    Float A = dst0.y * dst0.y + dst1.y * dst1.y + 1;
    Float B = -2 * (dst0.x * dst0.y + dst1.x * dst1.y);
    Float C = dst0.x * dst0.x + dst1.x * dst1.x + 1;
    Float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    Float det = -B * B + 4 * A * C;
    Float invDet = 1 / det;
    Float uSqrt = std::sqrt(det * C), vSqrt = std::sqrt(A * det);
    int s0 = (int)std::ceil(st.x - 2 * invDet * uSqrt), s1 = (int)std::floor(st.x + 2 * invDet * uSqrt);
    int t0 = (int)std::ceil(st.y - 2 * invDet * vSqrt), t1 = (int)std::floor(st.y + 2 * invDet * vSqrt);
    Spec sum(0.f);
    Float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        Float tt = it - st.y;
        for (int is = s0; is <= s1; ++is) {
            Float ss = is - st.x;
            Float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = smin((int)(r2 * 128), 128 - 1);
                Float weight = tx.weightLut[index];
                sum += MipTexel(tx, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return sum / sumWts;
}
inline Spec MipLookup(const Texture &tx, const P2 &st, P2 dst0, P2 dst1) {
    if (tx.trilinear) {
        Float width = smax(smax(std::abs(dst0.x), std::abs(dst0.y)), smax(std::abs(dst1.x), std::abs(dst1.y)));
        return MipLookupWidth(tx, st, 2 * width);
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
    Float majorLength = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
    Float minorLength = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minorLength * tx.maxAniso < majorLength && minorLength > 0) {
        Float scale = majorLength / (minorLength * tx.maxAniso);
        dst1.x *= scale; dst1.y *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return MipTriangle(tx, 0, st);
    const int nLevels = (int)tx.levels.size();
    Float lod = smax((Float)0, nLevels - (Float)1 + Log2F(minorLength));
    int ilod = (int)std::floor(lod);
    return LerpS(lod - ilod, MipEWA(tx, ilod, st, dst0, dst1), MipEWA(tx, ilod + 1, st, dst0, dst1));
}
// ImageTexture::Evaluate (textures/imagemap.h:82-89) over UVMapping2D::Map
inline Spec EvalImageTexture(const Texture &tx, const SurfaceInteraction &si) {
    P2 dstdx(tx.su * si.dudx, tx.sv * si.dvdx), dstdy(tx.su * si.dudy, tx.sv * si.dvdy);
    P2 st(tx.su * si.uv.x + tx.du, tx.sv * si.uv.y + tx.dv);
    return MipLookup(tx, st, dstdx, dstdy);
}

// materials/matte.cpp:45-62 (Lambert or OrenNayar), materials/plastic.cpp:45-70, materials/mirror.cpp:44-56 (constant or
// image textures on Kd / Ks)
inline void ComputeScatteringFunctions(const Scene &scene, const Material &m, const SurfaceInteraction &si, BSDF *bsdf) {
    bsdf->Init(si);
    const Spec Kd = m.KdTex >= 0 ? EvalImageTexture(scene.textures[m.KdTex], si) : Spec(m.Kd[0], m.Kd[1], m.Kd[2]);
    const Spec Ks = m.KsTex >= 0 ? EvalImageTexture(scene.textures[m.KsTex], si) : Spec(m.Ks[0], m.Ks[1], m.Ks[2]);
    if (m.type == MAT_UBER) {        // materials/uber.cpp:45-108 (constant e, opacity, Kr, Kt; roughness = uroughness, sigma = vroughness; no bump map)
        const Float e = m.eta;
        Spec op = (m.opTex >= 0 ? EvalImageTexture(scene.textures[m.opTex], si) : Spec(m.opacity[0], m.opacity[1], m.opacity[2])).Clamp();      // opacity->Evaluate(*si).Clamp(), :53
        Spec t = (-op + Spec(1.f)).Clamp();
        if (!t.IsBlack()) {
            bsdf->eta = 1.f;
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_SPECULAR_TRANSMISSION; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.R = t; b.etaA = 1.f; b.etaB = 1.f;
        } else bsdf->eta = e;
        Spec kd = op * Kd.Clamp();
        if (!kd.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_LAMBERT; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.R = kd;
        }
        Spec ks = op * Ks.Clamp();
        if (!ks.IsBlack()) {
            Float roughu = m.roughness, roughv = m.sigma;
            if (m.remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_MICROFACET; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = ks;
            b.dist.alphax = roughu; b.dist.alphay = roughv;
            b.frDielectric = true; b.frEtaI = 1.f; b.frEtaT = e;
        }
        Spec kr = op * Spec(m.Kr[0], m.Kr[1], m.Kr[2]).Clamp();
        if (!kr.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_SPECULAR_REFLECTION; b.type = BSDF_REFLECTION | BSDF_SPECULAR; b.R = kr;
            b.frDielectric = true; b.frEtaI = 1.f; b.frEtaT = e;
        }
        Spec kt = op * Spec(m.Kt[0], m.Kt[1], m.Kt[2]).Clamp();
        if (!kt.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_SPECULAR_TRANSMISSION; b.type = BSDF_TRANSMISSION | BSDF_SPECULAR; b.R = kt; b.etaA = 1.f; b.etaB = e;
        }
    } else if (m.type == MAT_GLASS) {       // materials/glass.cpp:44-95 (Kd = Kt, Ks = Kr, roughness = eta, sigma = uroughness, glassVRough = vroughness)
        bsdf->eta = m.roughness;
        const Float eta = m.roughness;
        Float urough = m.sigma, vrough = m.glassVRough;
        Spec R = Ks.Clamp(), T = Kd.Clamp();
        if (R.IsBlack() && T.IsBlack()) return;
        const bool isSpecular = urough == 0 && vrough == 0;
        if (isSpecular) {      // (allowMultipleLobes is true for the path integrator: one FresnelSpecular lobe)
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_FRESNEL_SPECULAR; b.type = BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
            b.R = R; b.S = T; b.etaA = 1.f; b.etaB = eta;
        } else {               // rough dielectric: MicrofacetReflection + MicrofacetTransmission over one Trowbridge-Reitz distribution (:66-93)
            if (m.remap) { urough = RoughnessToAlpha(urough); vrough = RoughnessToAlpha(vrough); }
            if (!R.IsBlack()) {
                BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
                b.kind = BXDF_MICROFACET; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = R;
                b.dist.alphax = urough; b.dist.alphay = vrough;
                b.frDielectric = true; b.frEtaI = 1.f; b.frEtaT = eta;
            }
            if (!T.IsBlack()) {
                BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
                b.kind = BXDF_MICROFACET_TRANSMISSION; b.type = BSDF_TRANSMISSION | BSDF_GLOSSY; b.R = T;
                b.dist.alphax = urough; b.dist.alphay = vrough; b.etaA = 1.f; b.etaB = eta;
            }
        }
    } else if (m.type == MAT_MIRROR) {      // materials/mirror.cpp:44-56 (Kr in Ks)
        Spec R = Ks.Clamp();
        if (!R.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_SPECULAR_REFLECTION; b.type = BSDF_REFLECTION | BSDF_SPECULAR; b.R = R;
        }
    } else if (m.type == MAT_MATTE) {
        Spec r = Kd.Clamp();
        Float sig = Clamp(m.sigma, 0, 90);
        if (!r.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.R = r;
            if (sig == 0) b.kind = BXDF_LAMBERT;
            else {      // OrenNayar::OrenNayar, core/reflection.h:414-420
                b.kind = BXDF_OREN_NAYAR;
                Float sigma = Radians(sig);
                Float sigma2 = sigma * sigma;
                b.A = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                b.B = 0.45f * sigma2 / (sigma2 + 0.09f);
            }
        }
    } else if (m.type == MAT_SUBSTRATE) {      // materials/substrate.cpp:44-65 (roughness = uroughness, sigma = vroughness)
        Spec d = Kd.Clamp(), sp = Ks.Clamp();
        Float roughu = m.roughness, roughv = m.sigma;
        if (!d.IsBlack() || !sp.IsBlack()) {
            if (m.remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_FRESNEL_BLEND; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = d; b.S = sp;
            b.dist.alphax = roughu; b.dist.alphay = roughv;
        }
    } else if (m.type == MAT_METAL) {          // materials/metal.cpp:59-79 (Kd = eta, Ks = k; no Clamp)
        Float uRough = m.roughness, vRough = m.sigma;
        if (m.remap) { uRough = RoughnessToAlpha(uRough); vRough = RoughnessToAlpha(vRough); }
        BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
        b.kind = BXDF_MICROFACET_CONDUCTOR; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = Spec(1.);
        b.S = Spec(m.Kd[0], m.Kd[1], m.Kd[2]); b.K = Spec(m.Ks[0], m.Ks[1], m.Ks[2]);
        b.dist.alphax = uRough; b.dist.alphay = vRough;
    } else {
        Spec kd = Kd.Clamp();
        if (!kd.IsBlack()) {
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_LAMBERT; b.type = BSDF_REFLECTION | BSDF_DIFFUSE; b.R = kd;
        }
        Spec ks = Ks.Clamp();
        if (!ks.IsBlack()) {
            Float rough = m.roughness;
            if (m.remap) rough = RoughnessToAlpha(rough);
            BxDF &b = bsdf->bxdfs[bsdf->nBxDFs++];
            b.kind = BXDF_MICROFACET; b.type = BSDF_REFLECTION | BSDF_GLOSSY; b.R = ks;
            b.dist.alphax = rough; b.dist.alphay = rough;
        }
    }
}

// Interaction subset used by lights (core/interaction.h:51-104)
struct Interaction { V3 p, pError, n; };

// Interaction::SpawnRay / SpawnRayTo, core/interaction.h:64-78
inline Ray SpawnRay(const V3 &p, const V3 &pError, const V3 &n, const V3 &d) {
    V3 o = OffsetRayOrigin(p, pError, n, d);
    return Ray(o, d, Infinity);
}
inline Ray SpawnRayTo(const Interaction &a, const Interaction &it) {
    V3 origin = OffsetRayOrigin(a.p, a.pError, a.n, it.p - a.p);
    V3 target = OffsetRayOrigin(it.p, it.pError, it.n, origin - it.p);
    V3 d = target - origin;
    return Ray(origin, d, 1 - ShadowEpsilon);
}

// Sphere::Sample(u), shapes/sphere.cpp:217-230
inline Interaction SphereSampleArea(const Sphere &s, bool reverseOrientation, const P2 &u, Float *pdf) {
    V3 pObj = V3(0, 0, 0) + s.radius * UniformSampleSphere(u);
    Interaction it;
    it.n = Normalize(XfNormal(s.w2o, V3(pObj.x, pObj.y, pObj.z)));
    if (reverseOrientation) it.n *= -1;
    pObj *= s.radius / Distance(pObj, V3(0, 0, 0));
    V3 pObjError = gamma(5) * Abs(pObj);
    it.p = XfPointErr2(s.o2w, pObj, pObjError, &it.pError);
    *pdf = 1 / (s.phiMax * s.radius * (s.zMax - s.zMin));   // 1/Area(), :215
    return it;
}
// Sphere::Sample(ref,u), shapes/sphere.cpp:232-292
inline Interaction SphereSample(const Sphere &s, bool reverseOrientation, const Interaction &ref, const P2 &u,
                                Float *pdf) {
    V3 pCenter = XfPoint(s.o2w, V3(0, 0, 0));
    V3 pOrigin = OffsetRayOrigin(ref.p, ref.pError, ref.n, pCenter - ref.p);
    if (DistanceSquared(pOrigin, pCenter) <= s.radius * s.radius) {
        Interaction intr = SphereSampleArea(s, reverseOrientation, u, pdf);
        V3 wi = intr.p - ref.p;
        if (wi.LengthSquared() == 0) *pdf = 0;
        else {
            wi = Normalize(wi);
            *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        }
        if (std::isinf(*pdf)) *pdf = 0.f;
        return intr;
    }
    V3 wc = Normalize(pCenter - ref.p);
    V3 wcX, wcY;
    CoordinateSystem(wc, &wcX, &wcY);
    Float sinThetaMax2 = s.radius * s.radius / DistanceSquared(ref.p, pCenter);
    Float cosThetaMax = std::sqrt(smax((Float)0, 1 - sinThetaMax2));
    Float cosTheta = (1 - u.x) + u.x * cosThetaMax;
    Float sinTheta = std::sqrt(smax((Float)0, 1 - cosTheta * cosTheta));
    Float phi = u.y * 2 * Pi;
    Float dc = Distance(ref.p, pCenter);
    Float ds = dc * cosTheta - std::sqrt(smax((Float)0, s.radius * s.radius - dc * dc * sinTheta * sinTheta));
    Float cosAlpha = (dc * dc + s.radius * s.radius - ds * ds) / (2 * dc * s.radius);
    Float sinAlpha = std::sqrt(smax((Float)0, 1 - cosAlpha * cosAlpha));
    V3 nWorld = SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc);
    V3 pWorld = pCenter + s.radius * V3(nWorld.x, nWorld.y, nWorld.z);
    Interaction it;
    it.p = pWorld;
    it.pError = gamma(5) * Abs(pWorld);
    it.n = nWorld;
    if (reverseOrientation) it.n *= -1;
    *pdf = 1 / (2 * Pi * (1 - cosThetaMax));
    return it;
}

// ---- triangle area lights: one DiffuseAreaLight per triangle of an emissive mesh (core/api.cpp:1609-1636) ----
// UniformSampleTriangle, core/sampling.cpp:154-157
inline P2 UniformSampleTriangle(const P2 &u) {
    Float su0 = std::sqrt(u.x);
    return P2(1 - su0, u.y * su0);
}
// Triangle::Area, shapes/triangle.cpp:576-582 (the literal 0.5 is a double)
inline Float TriangleArea(const V3 &p0, const V3 &p1, const V3 &p2) {
    return (Float)(0.5 * (double)Cross(p1 - p0, p2 - p0).Length());
}
// Triangle::Sample(u, pdf), shapes/triangle.cpp:596-621
inline Interaction TriangleSampleArea(const Mesh &mesh, const int *v, bool flip, const P2 &u, Float *pdf) {
    P2 b = UniformSampleTriangle(u);
    const V3 &p0 = mesh.p[v[0]], &p1 = mesh.p[v[1]], &p2 = mesh.p[v[2]];
    Interaction it;
    it.p = b.x * p0 + b.y * p1 + (1 - b.x - b.y) * p2;
    it.n = Normalize(Cross(p1 - p0, p2 - p0));
    if (mesh.hasN) {
        V3 ns(b.x * mesh.n[v[0]] + b.y * mesh.n[v[1]] + (1 - b.x - b.y) * mesh.n[v[2]]);
        it.n = Faceforward(it.n, ns);
    } else if (flip) it.n *= -1;
    V3 pAbsSum = Abs(b.x * p0) + Abs(b.y * p1) + Abs((1 - b.x - b.y) * p2);
    it.pError = gamma(6) * V3(pAbsSum.x, pAbsSum.y, pAbsSum.z);
    *pdf = 1 / TriangleArea(p0, p1, p2);
    return it;
}
// Shape::Sample(ref, u, pdf), core/shape.cpp:55-70
inline Interaction TriangleSample(const Mesh &mesh, const int *v, bool flip, const Interaction &ref, const P2 &u, Float *pdf) {
    Interaction intr = TriangleSampleArea(mesh, v, flip, u, pdf);
    V3 wi = intr.p - ref.p;
    if (wi.LengthSquared() == 0) *pdf = 0;
    else {
        wi = Normalize(wi);
        *pdf *= DistanceSquared(ref.p, intr.p) / AbsDot(intr.n, -wi);
        if (std::isinf(*pdf)) *pdf = 0.f;
    }
    return intr;
}

}  // namespace orc
