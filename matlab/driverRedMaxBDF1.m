function driverRedMaxBDF1(sceneID, batch)
%driverRedMaxBDF1  Same entry point and arguments as matlab-diff/driverRedMaxBDF1.m of the reference, time stepping on an MI355X.
%
%   driverRedMaxBDF1(sceneID, batch)     sceneID: a scene of scenesRedMax.m that the HIP path covers (0-9, 11, 14);
%                                        batch = true: no drawing, energies recorded, '### PASS ###' against Hexpected(1)
%
% Everything except the time stepping is the reference's own code (scenesRedMax, redmax.Scene and the joint / body
% classes); see redmax.runDriverHip.  Put this directory in front of matlab-diff/ on the MATLAB path.
if ~exist('sceneID', 'var') || isempty(sceneID), sceneID = 0; end
if ~exist('batch', 'var') || isempty(batch), batch = false; end
redmax.runDriverHip(sceneID, batch, 1);
end
