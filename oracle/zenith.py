"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, float64) of the cosine of the solar zenith angle that the reference's
loader appends as an input channel: `cos_zenith_angle(model_time, lon_grid, lat_grid)` at
/root/reference/utils/data_loader_era5.py:133-137, imported from `modulus.utils.zenith_angle` (:5).

The algorithm lives in a THIRD-PARTY dependency that is absent from /root/reference and from this image: NVIDIA Modulus
(`modulus/utils/zenith_angle.py`, Apache-2.0; the reference pins no version -- runtime image NGC PyTorch 23.07 + `pip install
nvidia-modulus`, README).  That file is itself the solar-position routine of FourCastNet-MIP / climt ("sunpos"): Greenwich
mean sidereal time from the IAU-82 polynomial, the Sun's ecliptic longitude from the mean anomaly / mean longitude series of
Meeus (Astronomical Algorithms, ch. 25, low-accuracy form), the obliquity polynomial of the ecliptic, then

    cos(zenith) = sin(lat) sin(dec) + cos(lat) cos(dec) cos(hour angle),   hour angle = local mean sidereal time - right ascension.

It is restated here from that published algorithm, function by function (names kept so it can be read side by side with the
upstream file).  PARITY UNPINNED at the dependency boundary: there is no copy of modulus to run and the reference's tests hold
no vectors for it; the restatement is pinned instead to hand-computable astronomical facts (tests/test_oracle_golden.py:
equinox / solstice declinations, the sub-solar point at the instants of the 2000 equinox and solstice, antipodal symmetry,
|equation of time| bounds).  The product path (swin_v2_weather_amd/utils/data_loader_era5.py::sun_position +
csrc/dataio.hip::era5_zenith_kernel) is a separate implementation that the tests hold to this one.
"""
import datetime

import numpy as np

RAD_PER_DEG = np.pi / 180.0
DATETIME_2000 = datetime.datetime(2000, 1, 1, 12, 0, 0)          # J2000.0 (UTC taken as UT1, like upstream)


def _days_from_2000(model_time):
    """days (float) since 2000-01-01 12:00"""
    dt = model_time - DATETIME_2000
    return dt.days + dt.seconds / 86400.0 + dt.microseconds / 86400.0e6


def _greenwich_mean_sidereal_time(model_time):
    """GMST in radians (IAU 1982 polynomial in Julian centuries from J2000, seconds of time -> /240 degrees).
    The cubic coefficient is written `6.2 * 10e-6` upstream (i.e. 6.2e-5, not the IAU value 6.2e-6); kept, it moves the
    angle by 4e-11 rad per century cubed."""
    jul_centuries = _days_from_2000(model_time) / 36525.0
    theta = 67310.54841 + jul_centuries * (876600 * 3600 + 8640184.812866 + jul_centuries * (0.093104 - jul_centuries * 6.2 * 10e-6))
    return np.deg2rad(theta / 240.0) % (2 * np.pi)


def _local_mean_sidereal_time(model_time, longitude):
    return _greenwich_mean_sidereal_time(model_time) + longitude


def _sun_ecliptic_longitude(model_time):
    """true ecliptic longitude of the Sun (radians): mean longitude + equation of centre"""
    jc = _days_from_2000(model_time) / 36525.0
    mean_anomaly = np.deg2rad(357.52910 + 35999.05030 * jc - 0.0001559 * jc * jc - 0.00000048 * jc * jc * jc)
    mean_longitude = np.deg2rad(280.46645 + 36000.76983 * jc + 0.0003032 * jc * jc)
    d_l = np.deg2rad((1.914600 - 0.004817 * jc - 0.000014 * jc * jc) * np.sin(mean_anomaly)
                     + (0.019993 - 0.000101 * jc) * np.sin(2 * mean_anomaly) + 0.000290 * np.sin(3 * mean_anomaly))
    return mean_longitude + d_l


def _obliquity_star(julian_centuries):
    """obliquity of the ecliptic (radians): 23 deg 26' 21.406" minus the polynomial in arc seconds"""
    jc = julian_centuries
    return np.deg2rad(23.0 + 26.0 / 60 + 21.406 / 3600.0
                      - (46.836769 * jc - 0.0001831 * jc ** 2 + 0.00200340 * jc ** 3 - 0.576e-6 * jc ** 4 - 4.34e-8 * jc ** 5) / 3600.0)


def _right_ascension_declination(model_time):
    jc = _days_from_2000(model_time) / 36525.0
    eps = _obliquity_star(jc)
    eclon = _sun_ecliptic_longitude(model_time)
    x = np.cos(eclon)
    y = np.cos(eps) * np.sin(eclon)
    z = np.sin(eps) * np.sin(eclon)
    r = np.sqrt(1.0 - z * z)
    declination = np.arctan2(z, r)
    right_ascension = 2 * np.arctan2(y, (x + r))
    return right_ascension, declination


def _local_hour_angle(model_time, longitude, right_ascension):
    return _local_mean_sidereal_time(model_time, longitude) - right_ascension


def _star_cos_zenith(model_time, lon, lat):
    """lon, lat in radians"""
    ra, dec = _right_ascension_declination(model_time)
    h_angle = _local_hour_angle(model_time, lon, ra)
    return np.sin(lat) * np.sin(dec) + np.cos(lat) * np.cos(dec) * np.cos(h_angle)


def cos_zenith_angle(time, lon, lat):
    """cos of the solar zenith angle; time: naive datetime (UTC), lon / lat in DEGREES (arrays broadcast)."""
    return _star_cos_zenith(time, np.deg2rad(np.asarray(lon, dtype=np.float64)), np.deg2rad(np.asarray(lat, dtype=np.float64)))


def era5_grids(H=721, W=1440):
    """the loader's grids (data_loader_era5.py:60-64: lat 90 .. -90 in 721 rows, lon 0 .. 359.75), cropped like the fields"""
    lat = np.linspace(90.0, -90.0, 721)[:H]
    lon = np.arange(0.0, 360.0, 0.25)[:W]
    return np.meshgrid(lon, lat)              # lon_grid, lat_grid  [H, W]
