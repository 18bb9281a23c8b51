"""
From a particle to the light source that stands for it: muon-like particles become a `MuonTrackLightSource`, everything that showers
a `ParticleCascadeLightSource` with L. Raedel's parametrisation of longitudinal profile, angular profile and light yield
(Raedel 2012; Raedel & Wiebusch arXiv:1206.5530, arXiv:1210.5140). Mirrors `theia.cascades` (src/theia/cascades.py): same names,
same numbers -- the fit coefficients are data -- and the same results, including two things a reader would not expect there:

* `createCascadeParameters` builds the angular parameters as `a = a_shift * logE + a_shift` and `b = b_slope * logE + a_shift`
  (cascades.py:186-187: the slope of `a` and the shift of `b` are never used);
* `createParamsFromParticle(x0=, density=)` hands both to converters that know them under other names (`X0`, `rho`), so the
  cascade is always parametrised for sea water (cascades.py:431-440, 512-517).

Both are kept so that a pipeline fed from this module traces what the reference would trace.
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import IntEnum
from math import log, log10, sqrt
from typing import Final

from scipy.stats import norm

from . import units as u
from .light import MuonTrackLightSource, ParticleCascadeLightSource

__all__ = ["CascadeLightYield", "CascadeParameters", "CascadePrimaryParticle", "EMinus", "EPlus", "Gamma", "K0_Long", "Neutron",
           "Particle", "ParticleType", "PiMinus", "PiPlus", "PMinus", "PPlus", "X0_ice", "X0_water", "createCascadeParameters",
           "createParamsFromParticle", "getCascadeParamsFromParticleType", "rho_ice", "rho_water"]


def __dir__():
    return __all__


X0_ice: Final[float] = 39.75 * u.cm    # radiation lengths
X0_water: Final[float] = 36.08 * u.cm
rho_ice: Final[float] = 0.91           # densities, g / cm^3
rho_water: Final[float] = 1.039        # (deep sea water)


class ParticleType(IntEnum):
    """PDG Monte Carlo particle numbers"""

    UNKNOWN = 0
    GAMMA = 22
    E_PLUS = -11
    E_MINUS = 11
    MU_PLUS = -13
    MU_MINUS = 13
    TAU_PLUS = -15
    TAU_MINUS = 15
    PI_0 = 111
    PI_PLUS = 211
    PI_MINUS = -211
    K0_LONG = 130
    NEUTRON = 2112
    P_PLUS = 2212
    P_MINUS = -2212


@dataclass
class Particle:
    particleType: ParticleType
    position: tuple[float, float, float]
    direction: tuple[float, float, float]
    energy: float
    startTime: float = 0.0
    length: float = float("NaN")  # track length where the particle has one
    speed: float = 1.0 * u.c


@dataclass
class CascadeLightYield:
    effectiveLength: float            # length of the Cherenkov track that gives as many photons
    effectiveLengthStd: float = 0.0   # its fluctuation (standard deviation)


@dataclass
class CascadeParameters:
    a_long: float
    b_long: float
    effectiveLength: float
    effectiveLengthStd: float = 0.0
    a_angular: float = 0.5375  # (average of the electromagnetic cascades)
    b_angular: float = 3.302


@dataclass(frozen=True)
class CascadePrimaryParticle:
    """fit coefficients of one kind of primary: each parameter is `shift + slope * log10(E)` or `alpha * E^beta`"""

    alpha_long: float
    beta_long: float
    b_long: float
    alpha_length: float = 5.321
    beta_length: float = 1.0
    alpha_length_std: float = 5.727e-2
    beta_length_std: float = 0.5
    a_angluar_shift: float = 0.5375  # (sic: the reference's spelling is part of its interface)
    a_angular_slope: float = 0.0
    b_angular_shift: float = 3.302
    b_angular_slope: float = 0.0


# name: (alpha_long, beta_long, b_long, alpha_length, beta_length, alpha_length_std, beta_length_std,
#        a_angular shift, slope, b_angular shift, slope) -- cascades.py:211-340
_FITS = {
    "EMinus": (2.01849, 1.45469, 0.63207, 5.3207078881, 1.00000211, 0.0578170887, 0.5, 0.53734995, 0.0, 3.30382993, 0.0),
    "EPlus": (2.00035, 1.45501, 0.63008, 5.3211320598, 0.99999254, 0.0573419669, 0.5, 0.5367158, 0.0, 3.30484209, 0.0),
    "Gamma": (2.83923, 1.45501, 0.64526, 5.3208540905, 0.99999877, 0.0566586567, 0.5, 0.53841841, 0.0, 3.29619817, 0.0),
    "PiPlus": (1.58357292, 0.96447937, 0.33833116, 3.3355182722, 1.03662217, 1.1920455395, 0.80772057,
               1.0299732199972658, -0.08806219920032332, 3.102713004779744, -0.12229465620485062),
    "PiMinus": (1.69176636, 0.93953506, 0.34108075, 3.3584489578, 1.03584394, 1.2250188073, 0.80322520,
                1.0412256610000645, -0.09187703681909758, 3.086039699134421, -0.11874011144663844),
    "K0_Long": (1.95948974, 0.80440041, 0.34535151, 3.2600450524, 1.03931457, 1.2141970572, 0.80779629,
                1.0591474180300977, -0.09635256670474648, 3.2258115113151793, -0.15816716921465757),
    "PPlus": (1.92249171, 0.77601150, 0.34969748, 2.8737183922, 1.05172118, 0.8804581378, 0.82445572,
              1.1574216500437113, -0.11090280215147694, 3.5079727644060794, -0.22892116764330248),
    "PMinus": (1.92249171, 0.77601150, 0.34969748, 3.0333074914, 1.04322206, 1.1323088104, 0.77134060,
               1.1574216500437113, -0.11090280215147694, 3.5079727644060794, -0.22892116764330248),
    "Neutron": (1.57739060, 0.93556570, 0.35269455, 2.7843854660, 1.05582906, 0.9322787137, 0.81776503,
                1.1292267334081203, -0.10876633838986713, 3.4157386880981093, -0.20638832466150736),
}
EMinus, EPlus, Gamma, PiPlus, PiMinus, K0_Long, PPlus, PMinus, Neutron = (CascadePrimaryParticle(*_FITS[k]) for k in (
    "EMinus", "EPlus", "Gamma", "PiPlus", "PiMinus", "K0_Long", "PPlus", "PMinus", "Neutron"))

_SHOWERING = {ParticleType.GAMMA: Gamma, ParticleType.E_MINUS: EMinus, ParticleType.E_PLUS: EPlus,
              ParticleType.PI_0: Gamma,  # (two photons at once)
              ParticleType.PI_PLUS: PiPlus, ParticleType.PI_MINUS: PiMinus, ParticleType.K0_LONG: K0_Long,
              ParticleType.P_PLUS: PPlus, ParticleType.P_MINUS: PMinus, ParticleType.NEUTRON: Neutron}
_TRACKS = {ParticleType.MU_PLUS, ParticleType.MU_MINUS, ParticleType.TAU_PLUS, ParticleType.TAU_MINUS}


def getCascadeParamsFromParticleType(t: ParticleType) -> CascadePrimaryParticle | None:
    """the fit of a primary, or `None` where there is none"""
    return _SHOWERING.get(t)


def createCascadeParameters(p: CascadePrimaryParticle, E: float, X0: float = X0_water, density: float = rho_water) -> CascadeParameters:
    """cascade of a primary `p` of energy `E` in a medium of radiation length `X0` and `density` (cascades.py:165-207)"""
    logE = max(0.0, log10(E))
    toMedium = 0.91 / density  # (the fits were made in ice: eq. 9 of arXiv:1210.5140)
    return CascadeParameters(
        a_long=p.alpha_long + p.beta_long * logE,
        b_long=X0 / p.b_long,
        effectiveLength=p.alpha_length * toMedium * E**p.beta_length,
        effectiveLengthStd=p.alpha_length_std * toMedium * E**p.beta_length_std,
        a_angular=p.a_angluar_shift * logE + p.a_angluar_shift,  # as the reference computes them: see the module's header
        b_angular=p.b_angular_slope * logE + p.a_angluar_shift,
    )


def _named(name: str, **params) -> dict:
    prefix = f"{name}__" if name else ""
    return {prefix + k: v for k, v in params.items()}


def _track(particle: Particle, name: str, uRand):
    if not particle.length > 0.0:  # (NaN as well)
        raise ValueError("particle is muon like, but no track length was specified!")
    d = particle.direction
    step = particle.length / sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    scale = 1.1880 + 0.0206 * log(particle.energy)  # secondaries below 500 MeV, arXiv:1206.5530
    length, std = particle.length * scale, sqrt(particle.length * 0.1 * u.m) * scale
    if uRand is not None:  # one draw of the yield; never less than the bare muon's
        length, std = max(length + norm.ppf(uRand).item() * std, particle.length), 0.0
    params = _named(name, startPosition=particle.position, startTime=particle.startTime,
                    endPosition=tuple(particle.position[k] + step * d[k] for k in range(3)),
                    endTime=particle.startTime + particle.length / particle.speed, muonEnergy=particle.energy)
    return MuonTrackLightSource, params, CascadeLightYield(length, std)


def _cascade(particle: Particle, name: str, uRand):
    fit = getCascadeParamsFromParticleType(particle.particleType)
    c = createCascadeParameters(fit, particle.energy, X0_water, rho_water)
    length, std = c.effectiveLength, c.effectiveLengthStd
    if uRand is not None:
        length, std = max(0.0, length + norm.ppf(uRand).item() * std), 0.0
    d = particle.direction
    norm_d = sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
    params = _named(name, startPosition=particle.position, startTime=particle.startTime, direction=(d[0] / norm_d, d[1] / norm_d, d[2] / norm_d),
                    effectiveLength=length, a_angular=c.a_angular, b_angular=c.b_angular, a_long=c.a_long, b_long=c.b_long)
    return ParticleCascadeLightSource, params, CascadeLightYield(length, std / length)


def createParamsFromParticle(particle: Particle, *, x0: float = X0_water, density: float = rho_water, lightSourceName: str = "lightSource",
                             uRand: float | None = None):
    """
    (light source class, its parameters, `CascadeLightYield`) for a particle. Parameter names carry `lightSourceName__` in front
    unless the name is empty; `uRand` in (0, 1) draws the light yield from its distribution instead of taking the mean. `x0` and
    `density` are accepted and, as in the reference, have no effect (see the module's header).
    """
    if particle.particleType in _TRACKS:
        return _track(particle, lightSourceName, uRand)
    if particle.particleType in _SHOWERING:
        return _cascade(particle, lightSourceName, uRand)
    raise ValueError(f"Could not create params from particle '{particle}'!")
